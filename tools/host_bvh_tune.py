#!/usr/bin/env python3
"""Host builder parameters (PRT_BVH_CI: SAH intersect cost; read from the environment by bvh.cpp) vs traversal rate.
  python tools/host_bvh_tune.py [--config C3] [--sets 1.5;1.0;2.0]"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3")
ap.add_argument("--sets", default="1.5;0.75;1.0;2.0;3.0;5.0")
a = ap.parse_args()
code = """
import sys, time
sys.path.insert(0, %r)
import torch, parallelraytracing_amd as prt
scene, cam, W, H, _, depth = prt.scenes.config(%r)
film = prt.Film(W, H); r = prt.HipWavefrontRenderer(device=0, max_depth=depth); r.Init(film, scene, cam)
info = r.bvh_info(); r.set_samples_in_flight(64); r.ProgressiveRender(64); r.synchronize(); r.reset_stats()
t0 = time.perf_counter(); r.render_async(64); r.render_async(64); r.synchronize(); dt = time.perf_counter() - t0
st = r.stats(); tr = r.measure_traversal()
print(f"build {info.build_ms:6.1f} ms nodes8 {info.n_nodes8} depth {info.depth8} {st.rays_total / dt / 1e6:8.1f} Mrays/s nodes/walked {tr.bvh_node_visits / max(1, tr.rays_traversed):.2f} tris/walked {tr.bvh_tri_tests / max(1, tr.rays_traversed):.2f} maxsp {tr.max_stack_used}")
""" % (ROOT, a.config)
for s in a.sets.split(";"):
    env = dict(os.environ, PRT_BVH_CI=s)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    out = [l for l in p.stdout.splitlines() if l.startswith("build")]
    print(f"CI {s:>5s}: {out[0] if out else 'FAILED ' + p.stderr[-300:]}", flush=True)
