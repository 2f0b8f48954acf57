#!/usr/bin/env python3
"""What one rank of an N-GPU run does, measured on ONE GPU: rank 0's tiles of the C3 frame for N = 1, 2, 4, 8 (the
image is tiled across the ranks, so the frame is fixed and every rank renders 1/N of the pixels at 256 spp per step).
step time x N / (N = 1 step time) = the strong-scaling efficiency to expect before the per-frame gather.
  python tools/sim_rank0.py [param=value ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import parallelraytracing_amd as prt  # noqa: E402

torch.cuda.set_device(0)
scene, cam, W, H, spp, depth = prt.scenes.config("C3")
params = dict(kv.split("=") for kv in sys.argv[1:])
base = None
for world in (1, 2, 4, 8):
    film = prt.Film(W, H)
    r = prt.HipWavefrontRenderer(device=0, max_depth=depth, rank=0, world_size=world)
    r.Init(film, scene, cam)
    for k, v in params.items():
        r.set_param(k, int(v))
    sif = 256  # bench.py's default at 1080p for every N
    r.set_samples_in_flight(sif)
    for _ in range(2):
        r.render_async(256)
    r.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        r.render_async(256)
    r.synchronize()
    dt = (time.perf_counter() - t0) / 4
    base = base or dt
    r.enable_timing(True)  # a second, instrumented pass: where the step goes (HIP events around every launch)
    r.reset_stats()
    for _ in range(4):
        r.render_async(256)
    r.synchronize()
    st = r.stats()
    stages = {k: round(getattr(st, k + "_ms") / 4, 2) for k in ("raygen", "intersect", "shade", "accumulate")}
    print(f"world {world}: rank-0 step {dt * 1e3:.2f} ms  efficiency {base / (dt * world):.3f}  sif {sif} {params} "
          f"kernels {stages} sum {sum(stages.values()):.2f} ms", flush=True)
    del r
