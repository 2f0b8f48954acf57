#!/usr/bin/env python3
"""The reference's own preset scenes (analytic primitives only) at main()'s frame size, 1920x1080, 20 segments
(src/main.cpp:96-97, src/backend/cpu/renderer.h:34): GPU rate, and the oracle = CPU restatement of the reference CPU
backend (linear scan over all primitives, recursive TraceRay) on the host cores for comparison.
  python tests/presets_rate.py [--cpu-rows 16]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpu-rows", type=int, default=24, help="rows of the frame the CPU oracle renders (0 = skip)")
    args = ap.parse_args()
    import torch
    import parallelraytracing_amd as prt
    torch.cuda.set_device(0)
    W, H = 1920, 1080
    for name in ("RANDOM_BALLS_LARGE", "RANDOM_BALLS_MEDIUM", "DEFAULT", "CORNELL"):
        scene = prt.Scene(name)
        cam = prt.Camera(width=W, height=H)
        line = f"{name}: {len(scene.primitives)} prims, 1080p, depth 20:"
        for prim_bvh in (1, 0):
            film = prt.Film(W, H)
            r = prt.HipWavefrontRenderer(device=0, max_depth=20)
            r.set_param("prim_bvh", prim_bvh)
            r.Init(film, scene, cam)
            r.set_samples_in_flight(16)
            r.render_async(16)
            r.synchronize()
            r.reset_stats()
            t0 = time.perf_counter()
            r.render_async(16)
            r.synchronize()
            dt = time.perf_counter() - t0
            st = r.stats()
            line += f"  GPU prim_bvh={prim_bvh}: {dt / 16 * 1e3:.2f} ms/spp {st.rays_total / dt / 1e6:.0f} Mrays/s;"
            del r
            if len(scene.primitives) <= 16:
                break
        if args.cpu_rows:
            from oracle import oracle as orc
            cores = min(len(os.sched_getaffinity(0)), 16)
            y0 = H // 2
            t0 = time.perf_counter()
            _, _, rays = orc.OracleScene(scene.desc()).render(cam.desc(), W, H, spp=1, max_depth=20, seed=0, iterative=False,
                                                              n_threads=cores, rect=(0, y0, W, y0 + args.cpu_rows))
            dt = time.perf_counter() - t0
            line += f"  CPU oracle ({cores} threads, {args.cpu_rows} rows): {rays / dt / 1e6:.2f} Mrays/s"
        print(line, flush=True)


if __name__ == "__main__":
    main()
