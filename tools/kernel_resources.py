#!/usr/bin/env python3
"""Register / scratch / LDS table of every kernel of one csrc/*.hip file, read from hipcc's own resource remarks
(-Rpass-analysis=kernel-resource-usage; no GPU needed).  `--check-spills PATTERN` exits non-zero when a kernel whose demangled
name contains PATTERN spills or uses scratch - tools/check_spills.sh runs that over the shipped small-M / wide instantiations
(scratch traffic is vmcnt-ordered and drains the operand prefetch).

usage: python tools/kernel_resources.py sdformerflow_amd/csrc/ms_smallm.hip [--check-spills smallm_kernel]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = "--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-unused-function -Wno-pass-failed".split()


def resources(src):
    cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + os.environ.get("SDF_EXTRA_FLAGS", "").split() + \
        ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.abspath(src), "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(src))).stderr
    rows = []
    for blk in err.split("Function Name: ")[1:]:
        name = blk.split()[0]
        def g(key):
            m = re.search(key + r": (\d+)", blk)
            return int(m.group(1)) if m else -1
        rows.append(dict(name=name, vgpr=g("VGPRs"), agpr=g("AGPRs"), spill=g("VGPRs Spill"), sspill=g("SGPRs Spill"),
                         scratch=g(r"ScratchSize \[bytes/lane\]"), occ=g(r"Occupancy \[waves/SIMD\]"), lds=g(r"LDS Size \[bytes/block\]")))
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.split("\n")
    for r, n in zip(rows, names):
        r["demangled"] = re.sub(r"^void sdfmm::\(anonymous namespace\)::", "", n).split("(")[0]
    return rows


def main():
    src = sys.argv[1]
    pat = sys.argv[sys.argv.index("--check-spills") + 1] if "--check-spills" in sys.argv else None
    rows = resources(src)
    bad = 0
    for r in rows:
        flag = ""
        if pat is not None and pat in r["demangled"] and (r["spill"] > 0 or r["scratch"] > 0):
            flag, bad = "   <-- SPILLS", bad + 1
        print(f"{r['demangled'][:100]:100s} vgpr {r['vgpr']:3d} agpr {r['agpr']:3d} spill {r['spill']:3d} scratch {r['scratch']:4d} B "
              f"occ {r['occ']} lds {r['lds']:6d}{flag}")
    if bad:
        print(f"{bad} kernel(s) matching '{pat}' spill", file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()
