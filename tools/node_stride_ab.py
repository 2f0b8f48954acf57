#!/usr/bin/env python3
"""A/B of the 8-wide node slot size (80-B packed vs one node per 128-B line): python tools/node_stride_ab.py --config C5 --spp 16"""
import argparse, os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C5"); ap.add_argument("--spp", type=int, default=16); ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--gpu-build", type=int, default=0)
args = ap.parse_args()
import numpy as np
import parallelraytracing_amd as prt
scene, cam, W, H, _, depth = prt.scenes.config(args.config)
rs, ref = {}, None
for stride in (5, 8):
    film = prt.Film(W, H)
    r = prt.HipWavefrontRenderer(device=0, max_depth=depth, seed=0)
    r.set_param("node_stride", stride)
    if args.gpu_build:
        r.set_param("gpu_build", args.gpu_build)
    r.Init(film, scene, cam)
    r.set_samples_in_flight(args.spp)
    r.ProgressiveRender(1)
    a = r.download().accum.copy()
    if ref is None:
        ref = a
    print(f"stride {stride}: image {'identical' if np.array_equal(a, ref) else 'DIFFERS'}", flush=True)
    r.render_async(args.spp); r.synchronize()
    rs[stride] = (r, film)
times = {5: [], 8: []}
for rd in range(args.rounds):
    for stride in (5, 8):
        r = rs[stride][0]
        r.reset_stats(); r.enable_timing(rd == args.rounds - 1)
        r.synchronize(); t0 = time.perf_counter(); r.render_async(args.spp); r.synchronize()
        times[stride].append(time.perf_counter() - t0)
        st = r.stats()
        if rd == args.rounds - 1:
            print(f"{args.config} stride {stride}: median {statistics.median(times[stride]) * 1e3:.2f} ms, {st.rays_total / statistics.median(times[stride]) / 1e6:.0f} Mrays/s, trav {st.intersect_ms:.2f} shade {st.shade_ms:.2f}", flush=True)
        r.enable_timing(False)
