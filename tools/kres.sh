#!/usr/bin/env bash
# register / LDS / spill table of every kernel of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage)
f=$1; shift
cd "$(dirname "$0")/../sdformerflow_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize "$@" -Rpass-analysis=kernel-resource-usage -c $f.hip -o /tmp/kres_$f.o 2>&1 | python3 -c "
import sys, re, subprocess
rows, cur = [], None
for ln in sys.stdin:
    m = re.search(r'remark:\s+(.*?)\s+\[-Rpass', ln)
    if not m: continue
    t = m.group(1)
    if t.startswith('Function Name:'):
        cur = {'name': t.split(':',1)[1].strip()}; rows.append(cur)
    elif cur is not None and ':' in t:
        k, v = t.split(':',1); cur[k.strip()] = v.strip()
for r in rows:
    nm = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip()
    nm = nm.replace('sdfmm::(anonymous namespace)::','').replace('(anonymous namespace)::','').replace('void ',''); nm = re.sub(r'\((sdfmm::)?[A-Za-z_:]*Params.*', '', nm)
    print(f\"{nm[:70]:70s} vgpr {r.get('VGPRs','?'):>4} agpr {r.get('AGPRs','?'):>3} spill {r.get('VGPRs Spill','?'):>3} scratch {r.get('ScratchSize [bytes/lane]','?'):>4} occ {r.get('Occupancy [waves/SIMD]','?'):>2} lds {r.get('LDS Size [bytes/block]','?'):>6}\")
"
