#!/usr/bin/env python3
"""Run the SNN en4 model at another BASELINE configuration and time it (no oracle: shapes, finiteness, determinism, ms).
usage: cfg_try.py B T H W [wh ww] [lif|psn]      e.g. config 5: cfg_try.py 4 20 480 640   large window: ... 15 15"""
import os, sys, time, torch, yaml
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from sdformerflow_amd.STSwinNet_SNN.Spiking_STSwinNet import MS_SpikingformerFlowNet_en4
from sdformerflow_amd.synthetic import synth_state_dict, synth_voxel
from sdformerflow_amd import harness
B, T, H, W = (int(v) for v in sys.argv[1:5])
wh, ww = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (9, 9)
kind = sys.argv[7] if len(sys.argv) > 7 else "lif"
cfg = yaml.safe_load(open(os.path.join(R, "sdformerflow_amd", "configs", "train_DSEC_supervised_SDformerFlow_en4.yml")))
cfg["model"]["spiking_neuron"] = dict(cfg["spiking_neuron"], neuron_type=kind, num_steps=T)
cfg["model"]["num_bins"] = T
cfg["swin_transformer"].update(input_size=[H, W], window_size=[2, wh, ww])
model = MS_SpikingformerFlowNet_en4(cfg["model"].copy(), cfg["swin_transformer"].copy())
model.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}), strict=True)
model = model.eval().to("cuda:0")
chunk = harness.prepare_chunk(synth_voxel(B, T, H, W, seed=1234 + 5)).to("cuda:0")
with torch.no_grad():
    a = [f.clone() for f in model(chunk)["flow"]]
    b = model(chunk)["flow"]
    torch.cuda.synchronize()
    assert all(f.shape == (B, 2, H, W) and torch.isfinite(f).all() for f in a)
    assert all(torch.equal(x, y) for x, y in zip(a, b)), "two runs differ"
    for _ in range(2): model(chunk)
    torch.cuda.synchronize(); t0 = time.time()
    n = 5
    for _ in range(n): model(chunk)
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
print(f"en4 {kind} B={B} T={T} {H}x{W} window (2,{wh},{ww}): {dt*1e3:.2f} ms per batch = {B/dt:.1f} samples/s; 4 finite flow maps, bit-equal "
      f"run to run; mean |flow| {a[-1].abs().mean().item():.3f}; peak memory {torch.cuda.max_memory_allocated()/2**30:.2f} GiB")
