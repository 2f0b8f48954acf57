#!/usr/bin/env python3
"""Turns the SQ / TCC / occupancy rocprofv3 passes of `bench.py` (tools/profile_round.sh) into profiles/<tag>_sq.json:
per launch of the dominant traversal kernel and of k_shade, the hardware's instruction counts and where the wave cycles
went.  Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave;
WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~= WAVE_CYCLES; SQ_INSTS_VALU counts wave-level instructions.

  python tools/pmc_sq.py <dir with pass_*/> <out.json> --config C3 --sif 256 --kernel k_traverse8_persistent [--jitter 0]
"""
import argparse
import collections
import csv
import glob
import json

ap = argparse.ArgumentParser()
ap.add_argument("root")
ap.add_argument("out")
ap.add_argument("--config", default="C3")
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--sif", type=int, default=256)
ap.add_argument("--jitter", type=int, default=0)
ap.add_argument("--kernel", default="k_traverse8_persistent")
args = ap.parse_args()


def targs(name):
    return [x.strip() for x in name[name.index("<") + 1:name.rindex(">")].split(",")] if "<" in name else []


def group_of(name):
    """'trav' for the non-instrumented instances of the dominant traversal kernel (the PRIM = true instance of bounce 0
    included: bench.py averages over all of them), 'shade' for k_shade, None for everything else."""
    base = name.split("(")[0].replace("void ", "")
    if args.kernel in base:
        a = targs(base)
        stats = (a[2] if "traverse8" in base else a[-1]) if a else "false"
        return None if stats == "true" else "trav"
    if base.startswith("k_shade"):
        return "shade"
    return None


sums = {"trav": collections.Counter(), "shade": collections.Counter()}
disp = {"trav": collections.defaultdict(set), "shade": collections.defaultdict(set)}
for f in sorted(glob.glob(args.root + "/pass_*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        g = group_of(r["Kernel_Name"])
        if g is None:
            continue
        sums[g][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[g][r["Counter_Name"]].add((f, r["Dispatch_Id"]))
dur = {"trav": [0.0, 0], "shade": [0.0, 0]}
for f in sorted(glob.glob(args.root + "/pass_sq/*/*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        g = group_of(r["Kernel_Name"])
        if g:
            dur[g][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            dur[g][1] += 1


def summarise(g):
    v, n = sums[g], {k: len(s) for k, s in disp[g].items()}
    if not v.get("SQ_WAVE_CYCLES"):
        return None
    per = lambda k: v[k] / max(1, n.get(k, 0))  # noqa: E731
    wc = v["SQ_WAVE_CYCLES"]
    out = {"launches": n["SQ_WAVE_CYCLES"],
           "valu_insts_per_launch": int(per("SQ_INSTS_VALU")),
           "wave_cycles_per_launch_quad": int(per("SQ_WAVE_CYCLES")),
           "waves_per_launch": int(per("SQ_WAVES")) if "SQ_WAVES" in v else None,
           "wave_cycles_share": {"active_valu": round(v["SQ_ACTIVE_INST_VALU"] / wc, 4),
                                 "active_any": round(v["SQ_ACTIVE_INST_ANY"] / wc, 4),
                                 "wait_inst_any": round(v["SQ_WAIT_INST_ANY"] / wc, 4),
                                 "wait_any": round(v["SQ_WAIT_ANY"] / wc, 4)},
           "profiled_avg_launch_ms": round(dur[g][0] / max(1, dur[g][1]), 4)}
    for extra in ("SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS",
                  "SQ_BUSY_CYCLES", "SQ_INSTS_SMEM", "SQ_LDS_BANK_CONFLICT"):
        if extra in v:
            out[extra.lower() + "_per_launch"] = int(per(extra))
    if "TCC_HIT_sum" in v:
        out["l2_hit_rate"] = round(v["TCC_HIT_sum"] / max(1.0, v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 4)
    if "MeanOccupancyPerCU" in v:
        out["mean_occupancy_per_cu"] = round(per("MeanOccupancyPerCU"), 2)
    if "GRBM_GUI_ACTIVE" in v and dur[g][1]:
        # rocprofv3 sums the 8 XCDs' counters; effective clock = cycles / 8 / time (MI355X_MICROARCH.md, DVFS give-back)
        out["effective_clock_GHz"] = round(per("GRBM_GUI_ACTIVE") / 8.0 / (dur[g][0] / dur[g][1] * 1e-3) / 1e9, 3)
    return out


t = summarise("trav")
assert t, "no counters found for " + args.kernel
res = {"config": args.config, "n_gpus": args.gpus, "samples_in_flight": args.sif, "jitter": args.jitter, "kernel": args.kernel}
res.update(t)
res["k_shade"] = summarise("shade")
res["note"] = ("separate rocprofv3 --pmc passes of bench.py (tools/profile_round.sh); quad-cycle counters summed over all waves; "
               "profiled launch times are longer than un-profiled ones (counter collection serialises dispatches)")
json.dump(res, open(args.out, "w"), indent=1)
print(open(args.out).read())
