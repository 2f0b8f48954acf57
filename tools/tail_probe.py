#!/usr/bin/env python3
"""Timeline of the traversal launches of one rank's share of a C3 step (instrumented kernel instance, s_memtime
stamps): when the first / last wave runs out of rays and how long the drain after that lasts.
  PRT_TAIL_PROBE=1 python tools/tail_probe.py [world] [spp] [param=value ...]"""
import os
import sys

os.environ["PRT_TAIL_PROBE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import parallelraytracing_amd as prt  # noqa: E402

torch.cuda.set_device(0)
scene, cam, W, H, spp, depth = prt.scenes.config("C3")
args = sys.argv[1:]
world = int(args.pop(0)) if args and args[0].isdigit() else 8
n = int(args.pop(0)) if args and args[0].isdigit() else 256
film = prt.Film(W, H)
r = prt.HipWavefrontRenderer(device=0, max_depth=depth, rank=0, world_size=world)
r.Init(film, scene, cam)
for kv in args:
    k, v = kv.split("=")
    r.set_param(k, int(v))
r.set_param("measure_spp", n)
r.set_samples_in_flight(n)
r.render_async(n)
r.synchronize()
for i in range(2):
    print(f"--- world {world}, {n} samples in the batch, run {i}", file=sys.stderr, flush=True)
    st = r.measure_traversal(0)
print("rays", st.rays_total, "traversed", st.rays_traversed)
