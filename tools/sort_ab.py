#!/usr/bin/env python3
"""Ray binning measured where it could matter (round-2 review, item 3): the 10 M-triangle configs, whose traversal is bound by
the memory system.  prt_set_param("sort_rays", m) makes every bounce >= 1 walk its front rays in the order of (Morton cell of
the origin in a 32^3 grid over the root box, direction octant) (m = 1) or (octant, cell) (m = 2), through a permutation
(nothing is moved; the sort itself is timed apart as scan_ms).  Prints per mode: traversal / sort / shade ms per batch,
node-loop and triangle-loop lane efficiencies, node visits per ray; the frames must be identical.
  python tools/sort_ab.py --config C5 --spp 64 --rounds 3"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C5")
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--jitter", type=int, default=0)
    args = ap.parse_args()
    import numpy as np
    import torch
    import parallelraytracing_amd as prt
    torch.cuda.set_device(0)
    scene, cam, W, H, spp_total, depth = prt.scenes.config(args.config)
    film = prt.Film(W, H)
    r = prt.HipWavefrontRenderer(device=0, max_depth=depth)
    r.Init(film, scene, cam)
    if args.jitter:
        r.set_sampling(jitter=1)
    r.set_samples_in_flight(args.spp)
    print(f"{args.config}: {scene.n_triangles} triangles, {W}x{H}, depth {depth}, {args.spp} samples per batch, kernel {r.kernel_instance()}", flush=True)
    ref = None
    res = {m: [] for m in (0, 1, 2)}
    for m in (0, 1, 2):
        r.set_param("sort_rays", m)
        film.Clear()
        r.frame_index = 0
        r.ProgressiveRender(2)
        a = r.download().accum.copy()
        if ref is None:
            ref = a
        print(f"mode {m}: frame identical to mode 0: {np.array_equal(a, ref)}", flush=True)
    for rd in range(args.rounds):
        for m in (0, 1, 2):
            r.set_param("sort_rays", m)
            r.reset_stats()
            r.enable_timing(True)
            r.render_async(args.spp)
            r.synchronize()
            st = r.stats()
            r.enable_timing(False)
            res[m].append((st.intersect_ms, st.scan_ms, st.shade_ms, int(st.rays_total)))
    for m in (0, 1, 2):
        r.set_param("sort_rays", m)
        r.set_param("measure_spp", min(64, args.spp))
        tr = r.measure_traversal()
        r.set_param("measure_spp", 1)
        t, s, sh = (statistics.median(x[i] for x in res[m]) for i in range(3))
        print(f"sort_rays={m}: traversal {t:8.3f} ms  sort {s:7.3f} ms  shade {sh:7.3f} ms per batch   "
              f"node-loop lane eff {tr.bvh_node_visits / max(1, tr.node_lane_slots):.3f}  tri-loop {tr.bvh_tri_tests / max(1, tr.tri_lane_slots):.3f}  "
              f"node visits / walked ray {tr.bvh_node_visits / max(1, tr.rays_traversed):.2f}", flush=True)
    r.set_param("sort_rays", 0)


if __name__ == "__main__":
    main()
