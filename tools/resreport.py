#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch table of prt_kernels.hip (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python3 tools/resreport.py [filter] [-- extra hipcc flags]
report(extra) returns the rows (tests/test_kernel_resources.py pins the occupancy-critical ones)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "parallelraytracing_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def report(extra=()):
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "--offload-arch=gfx950",
           "-c", os.path.join(CSRC, "prt_kernels.hip"), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + list(extra)
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = []
    names = []
    for ln in out.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = {"mangled": m.group(1)}
            rows.append(cur)
            names.append(m.group(1))
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r" SGPRs: (\d+)"),
                         ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, ln)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    if names:
        dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
        for r, d in zip(rows, dem):
            r["name"] = re.sub(r"^void ", "", re.sub(r"\(.*", "", d))
    return rows


if __name__ == "__main__":
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--")
        extra = args[i + 1:]
        args = args[:i]
    flt = args[0] if args else ""
    for r in report(extra):
        if flt in r.get("name", ""):
            print(f"{r['name']:75s} vgpr {r.get('vgpr')} agpr {r.get('agpr')} sgpr {r.get('sgpr')} scratch {r.get('scratch')} occ {r.get('occ')} lds {r.get('lds')}")
