#!/usr/bin/env python3
"""One rank's share of a config, a few steps (for traces): python tools/one_rank.py --world 8 --rank 0 --spp 256 --steps 4"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3"); ap.add_argument("--world", type=int, default=8); ap.add_argument("--rank", type=int, default=0)
ap.add_argument("--spp", type=int, default=256); ap.add_argument("--steps", type=int, default=4); ap.add_argument("--params", default="")
args = ap.parse_args()
import torch
import parallelraytracing_amd as prt
torch.cuda.set_device(0)
scene, cam, W, H, _, depth = prt.scenes.config(args.config)
film = prt.Film(W, H)
r = prt.HipWavefrontRenderer(device=0, max_depth=depth, seed=0, rank=args.rank, world_size=args.world)
r.Init(film, scene, cam)
for kv in filter(None, args.params.split(",")):
    k, v = kv.split("="); r.set_param(k, int(v))
r.set_samples_in_flight(min(args.spp, 256))
r.render_async(args.spp); r.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    r.render_async(args.spp)
r.synchronize()
dt = time.perf_counter() - t0
print(f"{args.steps} steps: {dt / args.steps * 1e3:.3f} ms per step")
