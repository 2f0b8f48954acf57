#!/usr/bin/env python3
"""Host (binned SAH + optimal collapse) vs the two device builders (gpu_build 1: PLOC + optimal collapse, 2: Morton octree) of the compressed 8-wide tree:
build time, tree size, and what the tree costs at traversal time.
  python tools/build_compare.py --config C3 --spp 32"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--spp", type=int, default=32)
    args = ap.parse_args()
    import torch
    import parallelraytracing_amd as prt
    torch.cuda.set_device(0)
    scene, cam, W, H, _, depth = prt.scenes.config(args.config)
    for gpu_build in (0, 1, 2):
        film = prt.Film(W, H)
        r = prt.HipWavefrontRenderer(device=0, max_depth=depth)
        r.set_param("gpu_build", gpu_build)
        t0 = time.perf_counter()
        try:
            r.Init(film, scene, cam)
        except prt.PrtError as e:
            print(f"{args.config} gpu_build={gpu_build}: {e}", flush=True)
            continue
        init_s = time.perf_counter() - t0
        info = r.bvh_info()
        r.set_samples_in_flight(args.spp)
        r.ProgressiveRender(args.spp)  # warm-up
        r.synchronize()
        r.reset_stats()
        t0 = time.perf_counter()
        r.render_async(args.spp)
        r.synchronize()
        dt = time.perf_counter() - t0
        st = r.stats()
        tr = r.measure_traversal()
        print(f"{args.config} gpu_build={gpu_build}: build {info.build_ms:8.1f} ms (Init incl. flatten/upload {init_s:.2f} s)  "
              f"nodes8 {info.n_nodes8}  depth {info.depth8}  {st.rays_total / dt / 1e6:8.1f} Mrays/s  "
              f"nodes/walked {tr.bvh_node_visits / max(1, tr.rays_traversed):.2f}  tris/walked {tr.bvh_tri_tests / max(1, tr.rays_traversed):.2f}",
              flush=True)
        del r


if __name__ == "__main__":
    main()
