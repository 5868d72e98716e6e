#!/usr/bin/env bash
# kernel-time breakdown of one steady-state configs[3] training step (local batch 4).  usage (GPU box): tools/prof_train.sh <out file>
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=${1:-gpurun_out/train_step.txt}
rm -rf /tmp/prof_train
rocprofv3 --kernel-trace -d /tmp/prof_train -o t --output-format csv -- python3 bench.py --train --local-batch 4 --steps 4 --warmup 2 > /tmp/prof_train.log 2>&1
MS=$(grep -o '"ms_per_step": [0-9.]*' /tmp/prof_train.log | grep -o '[0-9.]*$')
echo "# rocprofv3 --kernel-trace of: python3 bench.py --train --local-batch 4 --steps 4 --warmup 2; tools/prof_train_step.py on the last step ($MS ms under the profiler)" > $OUT
python3 tools/prof_train_step.py /tmp/prof_train $MS >> $OUT
