#!/usr/bin/env bash
# Timing-only ablation builds of the spike GEMM (run on the GPU box): which phase dominates?
set -e
cd "$(dirname "$0")/.."
for ab in 0 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -DSDF_ABLATE=$ab -c sdformerflow_amd/csrc/spike_gemm.hip -o /tmp/sg_$ab.o
  cp sdformerflow_amd/csrc/libsdformerflow_hip.so /tmp/lib_backup.so 2>/dev/null || true
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o sdformerflow_amd/csrc/libsdformerflow_hip.so /tmp/sg_$ab.o sdformerflow_amd/csrc/obj/neuron.o sdformerflow_amd/csrc/obj/qk_gate.o sdformerflow_amd/csrc/obj/elementwise.o sdformerflow_amd/csrc/obj/win_attn.o
  echo "=== SDF_ABLATE=$ab"
  python tools/gemm_microbench.py 2>&1 | grep -E "s0 fc1 f32|s0 q/k|s2 fc2|s3 fc2"
done
