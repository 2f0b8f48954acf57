#!/usr/bin/env python3
"""The reference's interactive contract: ONE sample per ProgressiveRender call (src/main.cpp render loop).  Calls per second
and rays per second at that granularity.  python tools/interactive_rate.py [--config C3] [--calls 100]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3"); ap.add_argument("--calls", type=int, default=100)
a = ap.parse_args()
import parallelraytracing_amd as prt
for cfg in a.config.split(","):
    scene, cam, W, H, _, depth = prt.scenes.config(cfg)
    film = prt.Film(W, H)
    r = prt.HipWavefrontRenderer(device=0, max_depth=depth, seed=0)
    r.Init(film, scene, cam)
    for kv in filter(None, os.environ.get("PRT_PARAMS", "").split(",")):  # A/B: PRT_PARAMS=name=value,...
        k, v = kv.split("=")
        r.set_param(k, int(v))
    for _ in range(5):
        r.ProgressiveRender()
    r.reset_stats()
    t0 = time.perf_counter()
    for _ in range(a.calls):
        r.ProgressiveRender()          # synchronous: returns when the sample is in the film
    dt = time.perf_counter() - t0
    st = r.stats()
    print(f"{cfg} {W}x{H} depth {depth}: {a.calls / dt:.0f} ProgressiveRender() calls per second ({dt / a.calls * 1e3:.2f} ms each), {st.rays_total / dt / 1e6:.0f} Mrays/s", flush=True)
    # where a call's time goes (HIP events around every launch; the events themselves add a little)
    r.reset_stats()
    r.enable_timing(True)
    for _ in range(a.calls):
        r.ProgressiveRender()
    st = r.stats()
    r.enable_timing(False)
    n = a.calls
    print(f"   per call, us: raygen {st.raygen_ms / n * 1e3:.0f}  traversal {st.intersect_ms / n * 1e3:.0f} ({st.intersect_launches / n:.0f} launches)  "
          f"shade {st.shade_ms / n * 1e3:.0f}  accumulate {st.accumulate_ms / n * 1e3:.0f}  rays per depth {[int(st.rays_per_depth[d] // n) for d in range(depth)]}", flush=True)
