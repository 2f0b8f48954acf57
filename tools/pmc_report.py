#!/usr/bin/env python3
"""Summarises the passes written by tools/pmc.sh: per kernel, counter sums over all dispatches and the
kernel's total duration.  usage: python tools/pmc_report.py gpurun_out/pmc2"""
import collections
import csv
import glob
import sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
ndisp = collections.Counter()
for f in sorted(glob.glob(root + "/pass*/*/*counter_collection.csv")):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (f, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
for f in sorted(glob.glob(root + "/pass1/*/*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        ndisp[k] += 1
for k in sorted(agg, key=lambda x: -dur[x]):
    if dur[k] < 0.05:
        continue
    v = agg[k]
    print(f"== {k}: {ndisp[k]} dispatches, {dur[k]:.2f} ms")
    for c in sorted(v):
        print(f"   {c:36s} {v[c]:.5g}")
    if "TCC_HIT_sum" in v:
        print(f"   -> L2 hit rate {v['TCC_HIT_sum'] / max(1, v['TCC_HIT_sum'] + v['TCC_MISS_sum']):.3f}")
    if "TCP_TCC_READ_REQ_sum" in v and v.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
        print(f"   -> L1 read miss ratio (TCC read req / cache accesses) {v['TCP_TCC_READ_REQ_sum'] / v['TCP_TOTAL_CACHE_ACCESSES_sum']:.3f}"
              f"   avg L1->L2 read latency {v['TCP_TCC_READ_REQ_LATENCY_sum'] / max(1, v['TCP_TCC_READ_REQ_sum']):.0f} cycles")
    if "TCP_UTCL1_TRANSLATION_MISS_sum" in v:
        print(f"   -> UTCL1 miss rate {v['TCP_UTCL1_TRANSLATION_MISS_sum'] / max(1, v['TCP_UTCL1_REQUEST_sum']):.4f}")
    if "FETCH_SIZE" in v:
        # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 reports half the bytes of wide coalesced reads (guide)
        print(f"   -> HBM-side read {v['FETCH_SIZE'] * 1024 / 1e6:.1f} MB (x2 if wide streaming), write {v.get('WRITE_SIZE', 0) * 1024 / 1e6:.1f} MB"
              f"  => {(v['FETCH_SIZE'] + v.get('WRITE_SIZE', 0)) * 1024 / 1e9 / (dur[k] / 1e3):.0f} GB/s uncorrected")
    if "SQ_WAVE_CYCLES" in v:
        wc = v["SQ_WAVE_CYCLES"]
        print(f"   -> of wave cycles: wait_any {v['SQ_WAIT_ANY'] / wc:.3f} wait_inst {v['SQ_WAIT_INST_ANY'] / wc:.3f} active_any {v['SQ_ACTIVE_INST_ANY'] / wc:.3f} "
              f"active_valu {v['SQ_ACTIVE_INST_VALU'] / wc:.3f};  VALU insts {v['SQ_INSTS_VALU']:.4g} VMEM_RD {v['SQ_INSTS_VMEM_RD']:.4g} LDS {v['SQ_INSTS_LDS']:.4g}")
