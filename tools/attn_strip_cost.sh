#!/usr/bin/env bash
# Does staging the relative-position bias (and shift mask) in LDS have anything to win?  Upper bound: a diagnostic build of the
# window-attention kernel whose bias / mask strip loads (16 x N floats per query tile, raw-buffer loads from L2 issued ahead of
# the K.Q^T MFMAs) are removed altogether (results wrong, timing only), against the product kernel.  GPU box.
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-pass-failed -DSDF_ATTN_NOSTRIP -c sdformerflow_amd/csrc/win_attn.hip -o /tmp/win_attn_nostrip.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libsdf_nostrip.so /tmp/win_attn_nostrip.o $(ls sdformerflow_amd/csrc/obj/*.o | grep -v win_attn)
for args in "ann 704 3 162 mask" "ann 704 3 162 nomask" "sew 704 3 162 mask" "ann 192 6 162 mask"; do
  echo "product:        $(python3 tools/win_attn_one.py $args | cut -c1-60)"
  echo "no strip loads: $(SDF_HIP_LIB=/tmp/libsdf_nostrip.so python3 tools/win_attn_one.py $args | cut -c1-60)"
done
