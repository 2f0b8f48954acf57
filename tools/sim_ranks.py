#!/usr/bin/env python3
"""Every rank's share of an N-GPU C3 step, one after the other on ONE GPU: step time, kernel time by stage and ray
count per rank.  Separates load imbalance between the ranks' tiles (ray counts differ) from fixed per-launch costs
(ray counts equal, times do not shrink with 1/N).
  python tools/sim_ranks.py [world] [rank=R] [param=value ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import parallelraytracing_amd as prt  # noqa: E402

torch.cuda.set_device(0)
scene, cam, W, H, spp, depth = prt.scenes.config("C3")
args = sys.argv[1:]
world = int(args.pop(0)) if args and args[0].isdigit() else 8
params = dict(kv.split("=") for kv in args)
only = params.pop("rank", None)  # rank=R: that rank only
for rank in ([int(only)] if only is not None else range(world)):
    film = prt.Film(W, H)
    r = prt.HipWavefrontRenderer(device=0, max_depth=depth, rank=rank, world_size=world)
    r.Init(film, scene, cam)
    for k, v in params.items():
        r.set_param(k, int(v))
    r.set_samples_in_flight(256)
    for _ in range(2):
        r.render_async(256)
    r.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        r.render_async(256)
    r.synchronize()
    dt = (time.perf_counter() - t0) / 4
    r.enable_timing(True)
    r.reset_stats()
    for _ in range(4):
        r.render_async(256)
    r.synchronize()
    st = r.stats()
    stages = {k: round(getattr(st, k + "_ms") / 4, 2) for k in ("raygen", "intersect", "shade", "accumulate")}
    per_depth = [st.rays_per_depth[d] // 4 for d in range(depth)]
    print(f"world {world} rank {rank}: step {dt * 1e3:.2f} ms  rays/step {st.rays_total // 4} traversed {st.rays_traversed // 4} "
          f"per depth {per_depth} kernels {stages} launches {st.intersect_launches // 4}", flush=True)
    del r
