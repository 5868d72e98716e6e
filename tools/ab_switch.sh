#!/usr/bin/env bash
# Same-box A/B of one diagnostic switch on the headline scheme: alternating runs of tools/batch_try.py with and without it.
# usage (GPU box): tools/ab_switch.sh "SDF_RES_MINC=96" [kind = lif] [R:F:steps = 4:2:960] [pairs = 3]
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
SW=$1; KIND=${2:-lif}; RF=${3:-4:2:960}; N=${4:-3}
for i in $(seq 1 $N); do
  echo "default   : $(python tools/batch_try.py $KIND $RF $RF 2>&1 | grep samples | tail -1)"
  echo "$SW : $(env $SW python tools/batch_try.py $KIND $RF $RF 2>&1 | grep samples | tail -1)"
done
