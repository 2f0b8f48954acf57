#!/usr/bin/env python3
"""Traversal rate vs the culling pad coefficient 2^-n: python tools/pad_ab.py --config C3 --spp 64 --pads 18,16,14,12"""
import argparse, os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3"); ap.add_argument("--spp", type=int, default=64); ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--pads", default="18,16,14,12")
args = ap.parse_args()
import numpy as np
import parallelraytracing_amd as prt
scene, cam, W, H, _, depth = prt.scenes.config(args.config)
rs, ref = {}, None
pads = [int(x) for x in args.pads.split(",")]
for pl in pads:
    film = prt.Film(W, H)
    r = prt.HipWavefrontRenderer(device=0, max_depth=depth, seed=0)
    r.set_param("pad_log2", pl)
    r.Init(film, scene, cam)
    r.set_samples_in_flight(args.spp)
    r.ProgressiveRender(1)
    a = r.download().accum.copy()
    if ref is None:
        ref = a
    print(f"pad 2^-{pl}: image {'identical' if np.array_equal(a, ref) else 'DIFFERS'}", flush=True)
    r.render_async(args.spp); r.synchronize()
    rs[pl] = r
times = {pl: [] for pl in pads}
for rd in range(args.rounds):
    for pl in pads:
        r = rs[pl]
        r.reset_stats(); r.enable_timing(rd == args.rounds - 1)
        r.synchronize(); t0 = time.perf_counter(); r.render_async(args.spp); r.synchronize()
        times[pl].append(time.perf_counter() - t0)
        st = r.stats()
        if rd == args.rounds - 1:
            print(f"{args.config} pad 2^-{pl}: median {statistics.median(times[pl]) * 1e3:.2f} ms, {st.rays_total / statistics.median(times[pl]) / 1e6:.0f} Mrays/s, trav {st.intersect_ms:.2f} shade {st.shade_ms:.2f}", flush=True)
        r.enable_timing(False)
