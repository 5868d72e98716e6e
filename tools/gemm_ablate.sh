#!/usr/bin/env bash
# Timing-only ablation builds of the spike GEMM (run on the GPU box): which phase dominates?
# Each variant is linked to its OWN library under /tmp and selected through SDF_HIP_LIB; the product library is not touched.
set -e
cd "$(dirname "$0")/.."
for ab in 0 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-pass-failed -DSDF_ABLATE=$ab -c sdformerflow_amd/csrc/spike_gemm.hip -o /tmp/sg_$ab.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libsdf_ablate_$ab.so /tmp/sg_$ab.o $(ls sdformerflow_amd/csrc/obj/*.o | grep -v '/spike_gemm\.o$')
  echo "=== SDF_ABLATE=$ab"
  SDF_HIP_LIB=/tmp/libsdf_ablate_$ab.so python tools/gemm_microbench.py 2>&1 | grep -E "s0 fc1 f32|s0 q/k|s2 fc2|s3 fc2"
done
