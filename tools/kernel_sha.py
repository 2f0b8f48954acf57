#!/usr/bin/env python3
"""Fingerprint of the kernel sources and the flags they are built with: sha256 over csrc/prt_kernels.hip, prt_device.h,
prt_kernels.h and the Makefile's HIPFLAGS line, first 16 hex digits.  tools/profile_round.sh stamps it into every
profiles/<tag>_{traffic,sq}.json, tools/isa_count.py into the ISA counts, and bench.py compares them with the sources it
runs on: a profile taken on other kernel code shows up in the bench line as "stale": true instead of passing silently."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "parallelraytracing_amd", "csrc")


def kernel_sha() -> str:
    h = hashlib.sha256()
    for f in ("prt_kernels.hip", "prt_device.h", "prt_kernels.h"):
        h.update(open(os.path.join(CSRC, f), "rb").read())
    for line in open(os.path.join(CSRC, "Makefile")):
        if line.startswith(("HIPFLAGS", "CXXFLAGS")):
            h.update(line.encode())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    if len(sys.argv) > 1:  # stamp JSON files in place
        import json
        for p in sys.argv[1:]:
            d = json.load(open(p))
            d["kernel_sha16"] = kernel_sha()
            json.dump(d, open(p, "w"), indent=1)
    print(kernel_sha())
