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
    ap.add_argument("--no-refit", action="store_true")
    args = ap.parse_args()
    import torch
    import parallelraytracing_amd as prt
    torch.cuda.set_device(0)
    scene, cam, W, H, _, depth = prt.scenes.config(args.config)
    for gpu_build, top in ((0, ""), (1, ""), (1, "device"), (2, "")):
        # PRT_PLOC_TOP=device: the quality builder clusters the top of the tree on the device too (full-search passes down to
        # the root) instead of the host's SAH sweep over the last <= 4096 clusters
        os.environ["PRT_PLOC_TOP"] = top
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
        print(f"{args.config} gpu_build={gpu_build}{' top on the device' if top else ''}: build {info.build_ms:8.1f} ms (Init incl. flatten/upload {init_s:.2f} s)  "
              f"nodes8 {info.n_nodes8}  depth {info.depth8}  {st.rays_total / dt / 1e6:8.1f} Mrays/s  "
              f"nodes/walked {tr.bvh_node_visits / max(1, tr.rays_traversed):.2f}  tris/walked {tr.bvh_tri_tests / max(1, tr.rays_traversed):.2f}",
              flush=True)
        del r
    os.environ["PRT_PLOC_TOP"] = ""
    if not args.no_refit:
        refit_rows(prt, args, scene, cam, W, H, depth)


def refit_rows(prt, args, scene, cam, W, H, depth):
    """prt_refit_meshes (SURVEY 8f-3 "refit"): the tree of the original mesh refitted on the device to a deformed copy
    (bend + per-vertex noise of `amp` x the mesh's unit size) against a fresh build of the deformed mesh: refit time and
    the traversal penalty (the refitted tree keeps the old topology)."""
    import numpy as np
    if len(scene.meshes) != 1 or scene.instances:
        return
    base, mat = scene.meshes[0]
    for gpu_build in (0, 1):
        for amp in (0.001, 0.005, 0.02):
            v = base.GetVertices().astype(np.float64)
            rng = np.random.default_rng(5)
            v[:, 0] += amp * 4.0 * np.sin(3.0 * v[:, 1])
            v[:, 2] += amp * 4.0 * np.cos(2.0 * v[:, 0])
            v += rng.normal(size=v.shape) * amp
            moved = prt.Mesh(vertices=v.astype(np.float32), normals=base.GetNormals(), indices=base.GetIndices())
            scene2 = prt.Scene(preset=None, sky=scene.sky)
            scene2.materials = list(scene.materials)
            scene2.primitives = list(scene.primitives)
            scene2.AddMesh(moved, mat)
            rates = {}
            for how in ("refit", "rebuild"):
                film = prt.Film(W, H)
                r = prt.HipWavefrontRenderer(device=0, max_depth=depth)
                r.set_param("gpu_build", gpu_build)
                r.Init(film, scene if how == "refit" else scene2, cam)
                t_refit = 0.0
                if how == "refit":
                    t0 = time.perf_counter()
                    r.Refit(scene2)
                    t_refit = time.perf_counter() - t0
                info = r.bvh_info()
                r.set_samples_in_flight(args.spp)
                r.ProgressiveRender(args.spp)
                r.synchronize()
                r.reset_stats()
                t0 = time.perf_counter()
                r.render_async(args.spp)
                r.synchronize()
                dt = time.perf_counter() - t0
                tr = r.measure_traversal()
                rates[how] = (r.stats().rays_total / dt / 1e6, tr.bvh_node_visits / max(1, tr.rays_traversed), info.refit_ms, t_refit * 1e3,
                              info.build_ms, r.kernel_instance())
                del r
            a, b = rates["refit"], rates["rebuild"]
            print(f"{args.config} gpu_build={gpu_build} deformation {amp}: refit {a[2]:.2f} ms on the device ({a[3]:.0f} ms incl. flatten + upload + read-back) "
                  f"vs rebuild {b[4]:.1f} ms;  refitted tree {a[0]:8.1f} Mrays/s ({a[1]:.2f} nodes/walked, {a[5]})  "
                  f"rebuilt tree {b[0]:8.1f} Mrays/s ({b[1]:.2f}, {b[5]})  penalty {100 * (1 - a[0] / b[0]):.1f} %", flush=True)


if __name__ == "__main__":
    main()
