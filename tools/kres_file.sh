#!/usr/bin/env bash
# Per-kernel register / LDS summary of one kernel file of sdformerflow_amd/csrc (hipcc cross-compiles without a GPU):
#   tools/kres_file.sh ms_wide
cd "$(dirname "$0")/../sdformerflow_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function -Wno-pass-failed \
  -Rpass-analysis=kernel-resource-usage -c "$1.hip" -o "/tmp/kres_$1.o" 2>&1 |
  python3 -c '
import re, sys
name = None
row = {}
for line in sys.stdin:
    if "error" in line or "warning:" in line:
        print(line.rstrip())
    m = re.search(r"remark: +(.*?): (.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        if name: print(name, row)
        name, row = v.replace("_ZN5sdfmm12_GLOBAL__N_1", ""), {}
    elif k in ("VGPRs", "AGPRs", "SGPRs", "VGPRs Spill", "ScratchSize [bytes/lane]", "LDS Size [bytes/block]", "Occupancy [waves/SIMD]"):
        row[k.split(" [")[0]] = v
if name: print(name, row)
'
