#!/usr/bin/env python3
"""Interleaved timing of parameter sets in ONE process.
  python tools/sweep.py --config C3 --sets "variant=0;variant=0,refill_min=1;variant=1" --rounds 5 --spp 8 --sif 4
Each set is a comma list of prt_set_param pairs (plus sif=N for samples in flight)."""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

DEFAULTS = {"variant": 0, "grid_blocks": 1024, "chunk": 256, "refill_min": 16, "exit_max": 16, "xcd_affinity": 0, "wide": 2, "stack_lds": 0, "tri_min": 0, "fuse": 0, "exact_grids": 1, "steal": 8, "tail": 1, "big": 2, "big_min": 96, "big_keep": 32, "static_small": 8}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--sets", required=True)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--spp", type=int, default=8)
    ap.add_argument("--sif", type=int, default=4)
    ap.add_argument("--eff", action="store_true", help="also report node-loop lane efficiency per set")
    args = ap.parse_args()
    import numpy as np
    import torch
    import parallelraytracing_amd as prt
    torch.cuda.set_device(0)
    scene, cam, W, H, spp_total, depth = prt.scenes.config(args.config)
    film = prt.Film(W, H)
    r = prt.HipWavefrontRenderer(device=0, max_depth=depth)
    r.Init(film, scene, cam)
    sets = []
    for sdesc in args.sets.split(";"):
        d = dict(DEFAULTS)
        d["sif"] = args.sif
        for kv in sdesc.split(","):
            if kv.strip():
                k, v = kv.split("=")
                d[k.strip()] = int(v)
        sets.append((sdesc, d))

    def apply(d):
        for k, v in d.items():
            if k == "sif":
                r.set_samples_in_flight(v)
            else:
                r.set_param(k, v)

    ref = None
    for name, d in sets:  # warm-up + bit-exactness check of every set against the first
        apply(d)
        film.Clear()
        r.frame_index = 0
        r.ProgressiveRender(2)
        a = r.download().accum.copy()
        if ref is None:
            ref = a
        elif not np.array_equal(a, ref):
            print(f"!! set [{name}] changes the image", flush=True)
        print(f"warm-up ok: [{name}]", flush=True)
    times = {name: [] for name, _ in sets}
    stages = {}
    for rd in range(args.rounds):
        for name, d in sets:
            apply(d)
            r.reset_stats()
            r.enable_timing(rd == args.rounds - 1)
            r.synchronize()
            t0 = time.perf_counter()
            r.render_async(args.spp)
            r.synchronize()
            dt = time.perf_counter() - t0
            st = r.stats()
            times[name].append((dt, st.rays_total))
            if rd == args.rounds - 1:
                stages[name] = (st.scan_ms, st.intersect_ms, st.shade_ms)
            r.enable_timing(False)
        print(f"round {rd} done", flush=True)
    for name, d in sets:
        ms = [t * 1e3 for t, _ in times[name]]
        rays = times[name][0][1]
        med = statistics.median(ms)
        line = (f"[{name:40s}] median {med:7.3f} ms  min {min(ms):7.3f}  {rays / med / 1e3:8.1f} Mrays/s  "
                f"scan {stages[name][0]:.2f} trav {stages[name][1]:.2f} shade {stages[name][2]:.2f}")
        if args.eff:
            apply(d)
            r.set_param("measure_spp", min(64, d["sif"]))
            tr = r.measure_traversal()
            r.set_param("measure_spp", 1)
            line += (f"  eff {tr.bvh_node_visits / max(1, tr.node_lane_slots):.3f} walked {tr.rays_traversed / tr.rays_total:.3f}"
                     f" tri-eff {tr.bvh_tri_tests / max(1, tr.tri_lane_slots):.3f} maxsp {tr.max_stack_used}"
                     f" nodes/walked {tr.bvh_node_visits / max(1, tr.rays_traversed):.2f} tris/walked {tr.bvh_tri_tests / max(1, tr.rays_traversed):.2f}")
            tot = max(1, tr.wave_cycles_refill + tr.wave_cycles_node + tr.wave_cycles_tri)
            line += (f"  wave time: refill {tr.wave_cycles_refill / tot:.2f} node {tr.wave_cycles_node / tot:.2f} tri {tr.wave_cycles_tri / tot:.2f}"
                     f"  cyc/node-iter {64 * tr.wave_cycles_node / max(1, tr.node_lane_slots):.0f} cyc/tri-iter {64 * tr.wave_cycles_tri / max(1, tr.tri_lane_slots):.0f}")
        print(line, flush=True)


if __name__ == "__main__":
    main()
