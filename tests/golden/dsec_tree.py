"""A tiny DSEC-preprocessed directory tree (layout of the reference's DSEC_dataset_lite.py) from seeded arrays.
Used by make_golden.py (read back through the REAL reference class) and by the test (read back through ours)."""
import os

import numpy as np

SEQS = {"zurich_city_00_a": 3, "thun_00_a": 2}
H, W, BINS = 12, 16, 5


def make_tree(root):
    rng = np.random.default_rng(20241218)
    for d in ("gt_tensors", "mask_tensors", "sequence_lists"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    single, double = [], []
    for seq, n in SEQS.items():
        names = [f"{seq}_{i:04d}.npy" for i in range(n)]
        for sub in ("05bins", "05bins_pol", "05frames"):
            os.makedirs(os.path.join(root, "event_tensors", sub, "left", seq), exist_ok=True)
        os.makedirs(os.path.join(root, "event_tensors", "01lists", "left"), exist_ok=True)
        for name in names:
            np.save(os.path.join(root, "gt_tensors", name), rng.normal(0, 5, (2, H, W)).astype(np.float32))
            np.save(os.path.join(root, "mask_tensors", name), rng.random((H, W)) < 0.7)
            vox = (rng.random((BINS, H, W)) < 0.1) * rng.uniform(-2, 2, (BINS, H, W))
            np.save(os.path.join(root, "event_tensors", "05bins", "left", seq, name), vox.astype(np.float32))
            np.save(os.path.join(root, "event_tensors", "05bins_pol", "left", seq, name), np.abs(vox).astype(np.float32))
            np.save(os.path.join(root, "event_tensors", "05frames", "left", seq, name),
                    rng.integers(0, 4, (BINS, 2, H, W)).astype(np.float32))
            ne = int(rng.integers(5, 20))
            ev = {"t": np.sort(rng.integers(0, 100000, ne)).astype(np.int64), "x": rng.integers(0, W, ne).astype(np.int16),
                  "y": rng.integers(0, H, ne).astype(np.int16), "p": rng.integers(0, 2, ne).astype(np.int8)}
            np.save(os.path.join(root, "event_tensors", "01lists", "left", name), np.array([ev], dtype=object), allow_pickle=True)
        single += names
        double += [(a, b) for a, b in zip(names[:-1], names[1:])]
    with open(os.path.join(root, "sequence_lists", "train_split_seq.csv"), "w") as f:
        f.write("".join(n + "\n" for n in single))
    with open(os.path.join(root, "sequence_lists", "train_split_doubleseq.csv"), "w") as f:
        f.write("".join(f"{a},{b}\n" for a, b in double))


def config(root, encoding="voxel", num_chunks=1, polarity=True, preprocessed=True):
    return {"data": {"path": root, "num_frames": BINS, "num_chunks": num_chunks, "preprocessed": preprocessed},
            "model": {"encoding": encoding}, "loader": {"resolution": [H, W], "polarity": polarity}}
