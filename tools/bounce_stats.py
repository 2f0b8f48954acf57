#!/usr/bin/env python3
"""Per-bounce work of the traversal kernel on a config: rays handed to it, node visits, triangle tests and active-lane
fractions of each bounce, from the instrumented instance (differences of runs with max_depth = 1, 2, ...: paths are
deterministic, so a run with max_depth d contains exactly the bounces 0 .. d-1 of the full run).
  python tools/bounce_stats.py [--config C3] [--spp 64] [--jitter 0] [--params name=value,...]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parallelraytracing_amd as prt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3")
ap.add_argument("--spp", type=int, default=64)
ap.add_argument("--jitter", type=int, default=0)
ap.add_argument("--params", default="")
a = ap.parse_args()
scene, cam, W, H, spp, depth = prt.scenes.config(a.config)
film = prt.Film(W, H)
r = prt.HipWavefrontRenderer(device=0, max_depth=depth, seed=0)
for kv in filter(None, a.params.split(",")):
    k, v = kv.split("=")
    r.set_param(k, int(v))
r.Init(film, scene, cam)
if a.jitter:
    r.set_sampling(jitter=1)
r.set_param("measure_spp", a.spp)
prev = None
print(f"{a.config}, {a.spp} samples in flight, jitter {a.jitter}: per bounce, per sample")
print("bounce | rays (all) | rays walked | node visits/walked ray | tri tests/walked ray | lane eff node | lane eff tri | node wave-steps | tri rounds")
for d in range(1, depth + 1):
    r.max_depth = d
    t = r.measure_traversal(sample=0)
    cur = dict(rays=int(t.rays_total), walked=int(t.rays_traversed), nodes=int(t.bvh_node_visits), tris=int(t.bvh_tri_tests),
               nslots=int(t.node_lane_slots), tslots=int(t.tri_lane_slots))
    dd = {k: cur[k] - (prev[k] if prev else 0) for k in cur}
    prev = cur
    w = max(1, dd["walked"])
    print(f"{d - 1} | {dd['rays'] / a.spp:.0f} | {dd['walked'] / a.spp:.0f} | {dd['nodes'] / w:.2f} | {dd['tris'] / w:.2f} | "
          f"{dd['nodes'] / max(1, dd['nslots']):.3f} | {dd['tris'] / max(1, dd['tslots']):.3f} | {dd['nslots'] / 64 / a.spp:.0f} | {dd['tslots'] / 64 / a.spp:.0f}")
