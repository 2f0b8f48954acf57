#!/usr/bin/env python3
"""One sample per ProgressiveRender call (the reference's contract), per frame size: the per-bounce pipeline (path_kernel = 0)
against the PATH instance (one launch that carries whole paths), the latter at several shade-step thresholds (refill_min).
  python tools/path_ab.py [--config C3] [--calls 60]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3"); ap.add_argument("--calls", type=int, default=60)
ap.add_argument("--sizes", default="1920x1080,1280x720,960x540,640x360,320x180")
ap.add_argument("--thresholds", default="16,32,48")
a = ap.parse_args()
import numpy as np
import parallelraytracing_amd as prt
scene, cam0, _, _, _, depth = prt.scenes.config(a.config)
for size in a.sizes.split(","):
    W, H = (int(v) for v in size.split("x"))
    cam = prt.Camera(position=tuple(cam0.position), front=tuple(cam0.front), width=W, height=H) if hasattr(cam0, "front") else prt.Camera(width=W, height=H)
    film = prt.Film(W, H)
    r = prt.HipWavefrontRenderer(device=0, max_depth=depth, seed=0)
    r.Init(film, scene, cam)
    r.set_param("path_max", 1 << 30)
    ref = None
    row = []
    for pk, thr in [(0, 16)] + [(1, int(t)) for t in a.thresholds.split(",")]:
        r.set_param("path_kernel", pk)
        r.set_param("refill_min", thr)
        film.Clear(); r.frame_index = 0
        for _ in range(3):
            r.ProgressiveRender()
        img = r.download().accum.copy()
        if ref is None:
            ref = img
        same = np.array_equal(img, ref)
        t0 = time.perf_counter()
        for _ in range(a.calls):
            r.ProgressiveRender()
        dt = (time.perf_counter() - t0) / a.calls
        row.append(f"{'pipeline' if pk == 0 else 'PATH thr ' + str(thr)}: {dt * 1e6:7.0f} us{'' if same else ' (FRAME DIFFERS)'}")
    r.set_param("refill_min", 16)
    print(f"{a.config} {W}x{H} ({W * H} paths per call): " + "   ".join(row), flush=True)
