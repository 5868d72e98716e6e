#!/usr/bin/env python3
"""SEW family (SpikingformerFlowNet, 3 encoders) forward at 1 x 10 x 2 x 288 x 384: ms per forward, and - under rocprofv3 --kernel-trace -
the evidence of which launches are still library kernels (tools/prof_sew.sh).  usage: sew_try.py [lif|psn] [forwards]"""
import sys
import time
import torch
import yaml
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sdformerflow_amd import harness
from sdformerflow_amd.STSwinNet_SNN.Spiking_STSwinNet import SpikingformerFlowNet
from sdformerflow_amd.synthetic import synth_state_dict, synth_voxel

kind = sys.argv[1] if len(sys.argv) > 1 else "lif"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg = yaml.safe_load(open(ROOT + "/sdformerflow_amd/configs/train_DSEC_supervised_SDformerFlow_en4.yml"))
cfg["model"]["spiking_neuron"] = dict(cfg["spiking_neuron"], neuron_type=kind)
cfg["swin_transformer"].update(input_size=[288, 384], swin_depths=[2, 2, 6], swin_num_heads=[3, 6, 12], swin_out_indices=[0, 1, 2])
model = SpikingformerFlowNet(cfg["model"].copy(), cfg["swin_transformer"].copy())
model.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}), strict=True)
model = model.eval().to("cuda:0")
x = harness.prepare_chunk(synth_voxel(1, 10, 288, 384, seed=5)).to("cuda:0")
for _ in range(3):
    out = model(x)["flow"]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    out = model(x)["flow"]
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print(f"SEW {kind} 1x10x2x288x384: {ms:.2f} ms per forward (eager, one stream); flows {[tuple(f.shape) for f in out]} finite {all(torch.isfinite(f).all().item() for f in out)}")
