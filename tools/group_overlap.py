#!/usr/bin/env python3
"""Does running the frame as K contexts on ONE GPU (K streams, K host threads, tiles split K ways) fill the launches'
ramps and tails?  python tools/group_overlap.py --config C3 --spp 64 --ks 1,2,4"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3"); ap.add_argument("--spp", type=int, default=64); ap.add_argument("--ks", default="1,2,4")
ap.add_argument("--reps", type=int, default=4); ap.add_argument("--jitter", type=int, default=0)
ap.add_argument("--params", default="", help="prt_set_param pairs for every context with K > 1, e.g. grid_blocks=512")
args = ap.parse_args()
import parallelraytracing_amd as prt
scene, cam, W, H, _, depth = prt.scenes.config(args.config)
for k in [int(x) for x in args.ks.split(",")]:
    film = prt.Film(W, H)
    r = prt.HipWavefrontGroupRenderer([0] * k, max_depth=depth, seed=0)
    r.Init(film, scene, cam)
    if args.jitter:
        r.set_sampling(jitter=1)
    for kv in filter(None, args.params.split(",")) if k > 1 else []:
        name, v = kv.split("=")
        r.set_param(name, int(v))
    r.set_samples_in_flight(min(args.spp, 256))
    r.ProgressiveRender(args.spp)
    ts = []
    r0 = r.stats().rays_total
    for _ in range(args.reps):
        t0 = time.perf_counter(); r.ProgressiveRender(args.spp); ts.append(time.perf_counter() - t0)
    rays = (r.stats().rays_total - r0) / args.reps
    print(f"{args.config} spp {args.spp} jitter {args.jitter}: {k} context(s) [{r.transport}]: best {min(ts) * 1e3:.2f} ms, median {sorted(ts)[len(ts) // 2] * 1e3:.2f} ms, {rays / min(ts) / 1e6:.0f} Mrays/s", flush=True)
    del r
