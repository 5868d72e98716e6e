#!/usr/bin/env bash
# The end-of-round measurement set in one GPU-box call: tools/final_measure.sh TAG -> gpurun_out/TAG_* (copied to profiles/ by hand).
#   1. counters per kernel over one launch sequence of the headline scheme (10 samples) -> TAG_pmc_forward.txt, TAG_kernel_counters.json
#      (copied to profiles/r6_kernel_counters.json BEFORE the bench so that its `roofline.traffic` / `by_kernel` roofs read this tree's counters)
#   2. the bench line (default command) and the driver's command shape (--steps 20 --warmup 5)
#   3. launch tables (library launch log): one sample, ten samples, psn
#   4. rocprofv3 kernel stats of the bench command and of config 3 (every kernel, untruncated), one training step
TAG=${1:-r6}
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
mkdir -p gpurun_out profiles
bash tools/pmc_forward2.sh ${TAG} lif 10 > gpurun_out/${TAG}_pmc_forward.txt 2>&1
cp gpurun_out/${TAG}_kernel_counters.json profiles/r6_kernel_counters.json
timeout 1500 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-sides --no-config3 --no-cpu > gpurun_out/${TAG}_bench_driver_shape.json 2>> gpurun_out/${TAG}_bench.err
for spec in "lif 1" "lif 10" "psn 10"; do
  set -- $spec
  python tools/launch_table.py $1 $2 seq > gpurun_out/${TAG}_launch_table_$1_R$2.txt 2>/dev/null
done
bash tools/prof_bench.sh ${TAG}b > /dev/null 2>&1
bash tools/prof_config3.sh ${TAG}c3 > gpurun_out/${TAG}_config3_kernel_stats.txt 2>&1
bash tools/prof_train.sh gpurun_out/${TAG}_train_step.txt > /dev/null 2>&1
rm -rf gpurun_out/prof_${TAG}b gpurun_out/prof_${TAG}c3
ls -la gpurun_out | grep ${TAG}
