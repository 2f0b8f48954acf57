#!/usr/bin/env python3
"""Turns two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of `bench.py` into profiles/<name>_traffic.json.
  python tools/pmc_traffic.py <fetch_pass_dir> <write_pass_dir> <out.json> --config C3 --spp-per-step 32 --sif 32
Unit and correction follow /opt/skills/guides (cdna_hip_programming.md §7, MI355X_MICROARCH.md §HBM):
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B, i.e. HALF the bytes of
wide (16 B/lane) reads, so the read side is doubled; WRITE_SIZE is exact for 16-B/lane stores.  The counters sit
on the L2's fabric side, so Infinity-Cache hits are INCLUDED: this is "bytes leaving L2", an upper bound on HBM."""
import argparse
import collections
import csv
import glob
import json

ap = argparse.ArgumentParser()
ap.add_argument("fetch_dir")
ap.add_argument("write_dir")
ap.add_argument("out")
ap.add_argument("--config", default="C3")
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--spp-per-step", type=int, default=32)
ap.add_argument("--sif", type=int, default=32)
ap.add_argument("--kernel", default="k_traverse8_persistent")
ap.add_argument("--jitter", type=int, default=0)
ap.add_argument("--read-factor", type=float, default=2.0,
                help="FETCH_SIZE correction: 2 for wide streaming reads (guide); what tools/gather_calib.hip measures for 80-B node gathers")
args = ap.parse_args()


def instrumented(name):
    """The STATS = true instances (prt_measure_traversal): k_traverse8_persistent<L, W, STATS, INST[, LEAN]>,
    k_traverse4_persistent<L, W, MODE, STATS>."""
    if "<" not in name:
        return False
    a = [x.strip() for x in name[name.index("<") + 1:name.rindex(">")].split(",")]
    return (a[2] if "traverse8" in name else a[-1]) == "true"


def family(name):
    """k_traverse8_persistent<L, W, STATS, INST, LEAN, PRIM>: the PRIM = true instance is the same kernel reading compact
    primary rays (bounce 0 of a batch); it belongs to the launches bench.py averages over."""
    if "traverse8" not in name or "<" not in name:
        return name
    a = [x.strip() for x in name[name.index("<") + 1:name.rindex(">")].split(",")]
    a += ["false"] * (5 - len(a))
    return name[:name.index("<") + 1] + ", ".join(a[:5]) + ">"


def total(d, counter):
    """Counter sum and dispatch count of the DOMINANT instance of the kernel (the template instance with the
    largest sum: the overflow-list re-traversal and the instrumented instance are separate, tiny dispatches)."""
    s, n = collections.Counter(), collections.defaultdict(set)
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            if args.kernel in name and r["Counter_Name"] == counter and not instrumented(name):
                name = family(name)
                s[name] += float(r["Counter_Value"])
                n[name].add(r["Dispatch_Id"])
    if not s:
        return 0.0, 0
    main = max(s, key=lambda k: s[k])
    return s[main], len(n[main])


fetch, n1 = total(args.fetch_dir, "FETCH_SIZE")
write, n2 = total(args.write_dir, "WRITE_SIZE")
assert n1 and n1 == n2, (n1, n2)
per_launch = (args.read_factor * fetch + write) * 1024.0 / n1
json.dump({"config": args.config, "n_gpus": args.gpus, "spp_per_step": args.spp_per_step, "samples_in_flight": args.sif,
           "jitter": args.jitter, "kernel": args.kernel, "launches": n1, "fetch_size_kib": fetch, "write_size_kib": write,
           "read_factor": args.read_factor, "hbm_bytes_per_launch": int(per_launch),
           "note": f"({args.read_factor:g} x FETCH_SIZE + WRITE_SIZE) x 1024 / launches; read factor = gfx950 FETCH_SIZE correction "
                   "(2 for wide streaming reads per the guide; the node / triangle gathers of this kernel calibrated with "
                   "tools/gather_calib.hip, profiles/r2_gather_calib.txt); fabric-side counter, includes Infinity-Cache hits"},
          open(args.out, "w"), indent=1)
print(open(args.out).read())
