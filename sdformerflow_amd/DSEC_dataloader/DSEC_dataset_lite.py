"""Reader for the reference's preprocessed DSEC tensors (`DSEC_dataloader/DSEC_dataset_lite.py:36-136`).

On-disk layout under `config["data"]["path"]` (written by the reference's `DSEC_dataset_preprocess.py`):

    gt_tensors/<seq>_<idx>.npy                flow label   (2, H, W) float
    mask_tensors/<seq>_<idx>.npy              valid mask   (H, W) bool
    event_tensors/<NN>bins/left/<seq>/<seq>_<idx>.npy        voxel, polarity kept in the sign    (NN, H, W)
    event_tensors/<NN>bins_pol/left/<seq>/<seq>_<idx>.npy    voxel, loader.polarity false
    event_tensors/<NN>frames/left/<seq>/<seq>_<idx>.npy      count frames (encoding "cnt")
    event_tensors/01lists/left/<seq>_<idx>.npy               raw event lists (data.preprocessed false)
    sequence_lists/<split>_split_seq.csv                     one file name per row              (num_chunks 1)
    sequence_lists/<split>_split_doubleseq.csv               two consecutive file names per row (num_chunks 2)

`__getitem__` returns `(chunk, mask, label)` exactly as the reference: with two chunks the voxels are concatenated on
the bin axis and mask / label belong to the SECOND file.  Only numpy / csv are needed (the reference's h5py / pandas /
tqdm imports are not used on this path).
"""
import csv
import os

import numpy as np
import torch
from torch.utils.data import Dataset


class DSECDatasetLite(Dataset):
    def __init__(self, config, file_list, stereo=False, transform=None, scale_factor=1):
        self.config = config
        root = config["data"]["path"]
        self.flow_path = os.path.join(root, "gt_tensors")
        self.mask_path = os.path.join(root, "mask_tensors")
        self.input = config["model"]["encoding"]
        self.num_frames_per_ts = config["data"]["num_frames"]
        self.num_chunks = config["data"]["num_chunks"]
        self.scale_factor = scale_factor
        self.height, self.width = int(config["loader"]["resolution"][0]), int(config["loader"]["resolution"][1])
        self.num_bins = self.num_frames_per_ts * self.num_chunks
        nn = str(self.num_frames_per_ts).zfill(2)
        if not config["data"]["preprocessed"]:
            self.events_path = os.path.join(root, "event_tensors", "01lists", "left")
        elif self.input == "voxel":
            sub = f"{nn}bins" if config["loader"]["polarity"] else f"{nn}bins_pol"
            self.events_path = os.path.join(root, "event_tensors", sub, "left")
        elif self.input == "cnt":
            self.events_path = os.path.join(root, "event_tensors", f"{nn}frames", "left")
        else:
            raise ValueError(f"unknown encoding {self.input!r}")
        if self.num_chunks not in (1, 2):
            raise AttributeError("num_chunks must be 1 or 2")
        suffix = "_split_doubleseq.csv" if self.num_chunks == 2 else "_split_seq.csv"
        with open(os.path.join(root, "sequence_lists", file_list + suffix), newline="") as f:
            self.files = [row for row in csv.reader(f) if row]
        self.transform = transform

    def __len__(self):
        return len(self.files)

    @staticmethod
    def _seq(name):
        return "_".join(name.split("_")[:-1])

    def __getitem__(self, idx):
        row = self.files[idx]
        first, target = row[0], row[self.num_chunks - 1]
        mask = torch.from_numpy(np.load(os.path.join(self.mask_path, target)))
        label = torch.from_numpy(np.load(os.path.join(self.flow_path, target)))
        if self.config["data"]["preprocessed"]:
            chunk = torch.from_numpy(np.load(os.path.join(self.events_path, self._seq(first), first), allow_pickle=True))
            if self.num_chunks == 2:
                second = torch.from_numpy(np.load(os.path.join(self.events_path, self._seq(target), target), allow_pickle=True))
                chunk = torch.cat((chunk, second), dim=0)
        else:
            ev = np.load(os.path.join(self.events_path, first), allow_pickle=True)
            chunk = {"ts": torch.from_numpy(ev[0]["t"]), "x": torch.from_numpy(ev[0]["x"]),
                     "y": torch.from_numpy(ev[0]["y"]), "p": torch.from_numpy(ev[0]["p"])}
        return chunk, mask, label
