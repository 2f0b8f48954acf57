#!/usr/bin/env python3
"""Kernel time vs gaps on the stream from a rocprofv3 kernel trace: python tools/trace_gaps.py <dir with *_kernel_trace.csv> [skip_first_n_raygen]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 1
# steps start at k_raygen
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_raygen")]
starts = starts[skip:]
for a, b in zip(starts[:-1], starts[1:]):
    seg = rows[a:b]
    wall = rows[b][0] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    gaps = [(seg[i + 1][0] - seg[i][1]) for i in range(len(seg) - 1)] + [rows[b][0] - seg[-1][1]]
    print(f"step of {len(seg)} kernels: wall {wall / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, gaps {sum(gaps) / 1e3:.1f} us (max {max(gaps) / 1e3:.1f})")
    if a == starts[len(starts) // 2]:
        for (s, e, n), g in zip(seg, gaps):
            print(f"    {n:48s} {(e - s) / 1e3:8.1f} us  then gap {g / 1e3:6.1f}")
