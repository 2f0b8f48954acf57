#!/usr/bin/env python3
"""How long does one launch of the traversal kernel take as a function of the number of rays?  Secondary (diffuse
bounce) rays of the C3 scene through the function-level entry point prt_closest_hit, n = 1 ... 1 M; run it under
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/floor -o f -- python3 tools/launch_floor.py
and read the k_traverse8_persistent durations in launch order (tools/launch_floor.py --parse <csv>)."""
import os
import sys

import numpy as np

SIZES = [1, 64, 64, 1024, 16384, 262144, 1048576, 64, 1]

if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    import csv
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    t = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "k_traverse8" in r["Kernel_Name"]]
    t = t[-len(SIZES):]
    for n, us in zip(SIZES, t):
        print(f"{n:8d} rays: {us:8.1f} us")
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import parallelraytracing_amd as prt  # noqa: E402

torch.cuda.set_device(0)
scene, cam, W, H, spp, depth = prt.scenes.config("C3")
film = prt.Film(W, H)
r = prt.HipWavefrontRenderer(device=0, max_depth=depth)
r.Init(film, scene, cam)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    r.set_param(k, int(v))
rng = np.random.default_rng(1)
px = rng.uniform(0, W, 1 << 21).astype(np.float32)
py = rng.uniform(0, H, 1 << 21).astype(np.float32)
o, d = r.camera_rays(px, py)
h = r.closest_hit(o, d)
on_mesh = h["prim"] >= 2  # C3: two analytic primitives (ground quad, light), then the triangles
pos, nrm = h["position"][on_mesh], h["normal"][on_mesh]
v = rng.normal(size=pos.shape).astype(np.float32)
v /= np.linalg.norm(v, axis=1, keepdims=True)
dirs = nrm + v
dirs /= np.maximum(np.linalg.norm(dirs, axis=1, keepdims=True), 1e-6)
print("secondary rays available:", pos.shape[0], flush=True)
for n in SIZES:
    hh = r.closest_hit(pos[:n], dirs[:n])
    print(n, "rays,", int((hh["prim"] >= 0).sum()), "hits", flush=True)
