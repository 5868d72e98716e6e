#!/usr/bin/env bash
# Build libsdformerflow_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# -fno-slp-vectorize: packed f32 VALU (v_pk_*) costs more than two scalar ops beside MFMAs and widens register pairs
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function -Wno-pass-failed ${SDF_EXTRA_FLAGS:-}"
mkdir -p obj
pids=()
for f in ms_wide ms_res ms_smallm neuron neuron_bwd bn_train qk_attn qk_front ms_mlp_fused pred_head pointwise_conv qk_gate_train spike_gemm spike_splitk switches launch_log spike_mm_pp spike_conv_wres spike_deconv_wres dense_conv_wres dense_linear linear_dw linear_train qk_gate elementwise win_attn ann_block ann_mlp_block head_tail; do
  if [ ! -f obj/$f.o ] || [ $f.hip -nt obj/$f.o ] || [ common.h -nt obj/$f.o ] || [ spike_mm.h -nt obj/$f.o ] || [ wide_common.h -nt obj/$f.o ] || [ switches.h -nt obj/$f.o ] || [ launch_log.h -nt obj/$f.o ] || [ ../../include/sdformerflow_hip.h -nt obj/$f.o ]; then
    # (hipcc's own per-kernel resource remarks are kept beside the object: tools/check_spills.py reads them - no second compile)
    ( $HIPCC $FLAGS -Rpass-analysis=kernel-resource-usage -c $f.hip -o obj/$f.o 2> obj/$f.res || { grep -v "Rpass-analysis\|^ *[0-9]* |\|^ *| *\^" obj/$f.res >&2; exit 1; }
      grep -A3 "warning:\|error:" obj/$f.res >&2 || true ) &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
# the kernels that hold whole compute units must not touch scratch (its traffic is vmcnt-ordered and drains the operand prefetch)
python3 ../../tools/check_spills.py obj/ms_smallm.res obj/ms_wide.res obj/ms_res.res obj/linear_dw.res obj/linear_train.res obj/ann_mlp_block.res
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libsdformerflow_hip.so obj/*.o
echo "built $(pwd)/libsdformerflow_hip.so"
