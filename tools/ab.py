#!/usr/bin/env python3
"""A/B timing of traversal-kernel variants in ONE process, interleaved rounds (cdna_hip_programming.md rule 24).
  python tools/ab.py --config C3 --variants 0,1 --rounds 5 --spp 8 [--sif 4]
Prints per variant: median / min ms per step, Mrays/s, and the HIP-event stage split."""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--variants", default="0,1")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--spp", type=int, default=8)
    ap.add_argument("--sif", type=int, default=4)
    ap.add_argument("--check", action="store_true", help="compare the films of all variants bit for bit")
    args = ap.parse_args()
    import numpy as np
    import torch
    import parallelraytracing_amd as prt
    torch.cuda.set_device(0)
    scene, cam, W, H, spp_total, depth = prt.scenes.config(args.config)
    variants = [int(v) for v in args.variants.split(",")]
    film = prt.Film(W, H)
    r = prt.HipWavefrontRenderer(device=0, max_depth=depth)
    r.Init(film, scene, cam)
    r.set_samples_in_flight(args.sif)
    b = r.bvh_info()
    print(f"{args.config}: {scene.n_triangles} tris, {b.n_nodes} nodes, depth {b.max_depth}, sah {b.sah_cost:.1f}")
    times = {v: [] for v in variants}
    stages = {}
    films = {}
    for v in variants:  # warm-up + optional correctness check
        r.set_variant(v)
        film.Clear()
        r.frame_index = 0
        r.ProgressiveRender(args.sif)
        if args.check:
            films[v] = r.download().accum.copy()
    if args.check:
        ref = films[variants[0]]
        for v in variants[1:]:
            print(f"variant {v} == variant {variants[0]}: {np.array_equal(films[v], ref)}")
    for rd in range(args.rounds):
        for v in variants:
            r.set_variant(v)
            r.reset_stats()
            r.enable_timing(rd == args.rounds - 1)
            r.synchronize()
            t0 = time.perf_counter()
            r.render_async(args.spp)
            r.synchronize()
            dt = time.perf_counter() - t0
            st = r.stats()
            times[v].append((dt, st.rays_total))
            if rd == args.rounds - 1:
                stages[v] = (st.raygen_ms, st.intersect_ms, st.shade_ms, st.accumulate_ms)
            r.enable_timing(False)
    for v in variants:
        ms = [t * 1e3 for t, _ in times[v]]
        rays = times[v][0][1]
        med = statistics.median(ms)
        print(f"variant {v}: median {med:.3f} ms  min {min(ms):.3f} ms per {args.spp} spp  -> {rays / med / 1e3:.1f} Mrays/s"
              f"   stages(ms) raygen {stages[v][0]:.2f} intersect {stages[v][1]:.2f} shade {stages[v][2]:.2f} acc {stages[v][3]:.2f}")
    tr = r.measure_traversal()
    r.set_variant(variants[0])
    tr = r.measure_traversal()
    eff = tr.bvh_node_visits / max(1, tr.node_lane_slots)
    print(f"variant {variants[0]} node-loop lane efficiency {eff:.3f} (visits / 64*wave iterations)")
    print(f"per sample: rays {tr.rays_total}  node visits/ray {tr.bvh_node_visits / tr.rays_total:.2f}  "
          f"tri tests/ray {tr.bvh_tri_tests / tr.rays_total:.2f}  rays/depth {[tr.rays_per_depth[d] for d in range(depth)]}")


if __name__ == "__main__":
    main()
