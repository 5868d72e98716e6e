#!/usr/bin/env bash
# The end-of-round measurement set in one GPU-box call: tools/final_measure.sh TAG -> gpurun_out/TAG_* (copied to profiles/ by hand)
#   bench line (default command), rocprofv3 kernel stats of that command, launch sequence of one forward, PMC per kernel over one
#   forward, PMC of the dense convolution (config 3 traffic), stamps of the small-M convolution.
TAG=${1:-r4}
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
timeout 1300 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
bash tools/prof_bench.sh ${TAG}b > /dev/null 2>&1
bash tools/prof_forward_one.sh ${TAG}s > /dev/null 2>&1
bash tools/pmc_forward2.sh ${TAG} > gpurun_out/${TAG}_pmc_forward.txt 2>&1
bash tools/pmc_dense_conv.sh > gpurun_out/${TAG}_pmc_dense_conv.txt 2>&1
[ -f build/smallm/libstamp.so ] && bash tools/smallm_ablate.sh stamp 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_smallm_stamps.txt
ls -la gpurun_out | grep ${TAG}
