#!/usr/bin/env python3
"""bench.py — headline benchmark: Mrays/s of the wavefront path tracer on BASELINE.json's metric config.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (config.workload): C3 of SURVEY.md §8(d) — the bundled dragon.ply refined by deterministic
longest-edge bisection to 870,000 triangles, on a ground quad under an emissive quad, 1920x1080,
max_depth = 5 segments ("4 bounces"), synthetic (there is no 1M-triangle scene in the reference).
A "step" is one complete frame of the config: all its samples per pixel (C3: 256 spp; --spp-per-step overrides)
over the whole image, followed by the per-frame gather of the ranks' tiles to rank 0 and the un-tiling into the
Film layout.  The frame is FIXED as N grows (the image is tiled across the GPUs), so scaling is "strong"; every
GPU keeps up to ~530 M paths in flight (min(spp, 256 x N) samples of its 1/N of the pixels; 68 GB of path state).
Rays = ray segments for which a closest-hit query ran, counted on the device.
The scene, BVH and path state are resident in HBM before the timed region starts.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
NODE_BYTES = 80        # one compressed 8-wide node visit: origin + exponents + 8 x (6 quantized planes + meta) (csrc/bvh.h)
NODE_BYTES4 = 128      # one BVH4 node visit (A/B kernels): four child AABBs + four child refs = one cache line
TRI_BYTES = 48         # one leaf triangle test: 3 x float4 (P0+prim, P1+material, P2)
RAY_FIXED_BYTES = 44   # traversal kernel per ray it walks: origin+dir (2 x 16 B) + hit id and d2 read (8 B) + hit id write (4 B)
PRIM_BYTES = 112       # one analytic primitive record (DevPrim)


MESH_OF = {"C2": "bunny.ply refined by longest-edge bisection", "C3": "dragon.ply refined by longest-edge bisection",
           "C4": "dragon.ply refined by longest-edge bisection", "C5": "12 baked copies of the refined dragon.ply",
           "C5I": "12 placed copies (PrtInstance, two-level BVH) of the refined dragon.ply"}


def load_traffic(config, world, spp_step, sif, kernel):
    import glob
    best = (None, "no committed PMC profile matches this configuration")
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        # a launch = one bounce of one batch of `sif` samples over this rank's pixels, whatever the step length
        if (t.get("config"), t.get("n_gpus"), t.get("samples_in_flight"), t.get("kernel")) == (config, world, sif, kernel):
            best = (t.get("hbm_bytes_per_launch"), f"{os.path.basename(f)}: {t.get('note', '')}")
    return best


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C3", help="C2 | C3 | C4 | C5 | C5I (SURVEY.md §8d; C5I = C5 as placed copies of one mesh)")
    ap.add_argument("--spp-per-step", type=int, default=0, help="0 = the config's full sample count (C3: 256)")
    ap.add_argument("--samples-in-flight", type=int, default=0, help="0 = auto")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--wide", type=int, default=2, help="2 = compressed 8-wide tree (default), 1 = 4-wide tree (A/B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL over xGMI) | gloo (rehearsal on a box with fewer GPUs)")
    ap.add_argument("--dump", default="", help="write the final frame as PPM/PFM with this path prefix")
    return ap.parse_args()


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    import __graft_entry__ as g
    g.build(quiet=True)
    import parallelraytracing_amd as prt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP kernels are the only compute path")
    if os.environ.get("PRT_BENCH_SAME_DEVICE"):  # rehearsal: all ranks share GPU 0 (use with --backend gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group(args.backend)

    # ---- workload (outside the timed region: PLY parse, refinement, BVH build, upload) ----
    t_setup = time.time()
    scene, cam, W, H, spp_total, max_depth = prt.scenes.config(args.config)
    film = prt.Film(W, H)
    r = prt.HipWavefrontRenderer(device=local_rank, max_depth=max_depth, seed=0, rank=rank, world_size=world)
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    r.Init(film, scene, cam)
    r.set_variant(args.variant)
    r.set_param("wide", args.wide)
    if os.environ.get("PRT_FUSE"):  # A/B: 0 = the producers store every ray (no fused analytic segment)
        r.set_param("fuse", int(os.environ["PRT_FUSE"]))
    n_tris = scene.n_triangles
    bvh = r.bvh_info()
    spp_step = args.spp_per_step or spp_total  # a step = one complete frame of the config
    # many samples in flight: the late bounces are latency/tail-bound, more rays per launch hide it (measured on C3 with
    # the current kernels: 64 -> 13.5, 128 -> 14.2, 256 -> 14.3 Grays/s); 288 GB of HBM make 68 GB of path state cheap
    sif = args.samples_in_flight or min(spp_step, max(1, (256 * 1920 * 1080 * world) // (W * H)))
    sif = max(1, min(sif, spp_step))
    r.set_samples_in_flight(sif)
    gather = prt.dist.FilmGather(r, device)
    setup_s = time.time() - t_setup

    # algorithmic traffic of the dominant kernel, measured with the instrumented traversal on the first samples
    # (a batch of up to 64 samples, like the timed batches: with compact primary rays a wave of bounce 0 holds 64 samples
    # of one pixel, which a one-sample batch cannot show in the active-lane fractions; counts are per sample below)
    n_meas = max(1, min(64, sif))
    r.set_param("measure_spp", n_meas)
    trav = r.measure_traversal(sample=0)
    rays_sample = int(trav.rays_total) // n_meas
    rays_walked = int(trav.rays_traversed) // n_meas  # rays that enter the BVH root box; the others never reach this kernel
    wide8 = int(bvh.n_nodes8) > 0 and args.wide == 2
    node_bytes = NODE_BYTES if wide8 else NODE_BYTES4
    kernel_name = "k_traverse8_persistent" if wide8 else "k_traverse4_persistent"
    node_visits_sample = int(trav.bvh_node_visits) // n_meas
    tri_tests_sample = int(trav.bvh_tri_tests) // n_meas
    alg_bytes_sample = node_bytes * node_visits_sample + TRI_BYTES * tri_tests_sample + RAY_FIXED_BYTES * rays_walked

    def step():
        r.render_async(spp_step)
        gather()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    r.reset_stats()
    r.enable_timing(not args.no_kernel_timing)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    r.enable_timing(False)
    st = r.stats()
    rays_local = int(st.rays_total)

    if world > 1:
        rdev = device if args.backend == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        c = torch.tensor([rays_local], dtype=torch.int64, device=rdev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        rays_total = int(c.item())
    else:
        rays_total = rays_local

    value = rays_total / dt / 1e6

    # ---- roofline of the dominant kernel (k_intersect), this rank ----
    roofline = None
    if not args.no_kernel_timing and st.intersect_launches:
        avg_ms = st.intersect_ms / st.intersect_launches
        # measure_traversal ran on THIS rank's tiles, so alg_bytes_sample is already the local share of one
        # sample; a launch is one depth of one batch of `sif` samples: bytes/launch = total bytes / launches
        bytes_per_launch = alg_bytes_sample * st.samples / st.intersect_launches
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        # measured HBM-side bytes per launch come from separate rocprofv3 --pmc passes of this same command
        # (profiles/*_traffic.json, written by tools/pmc_traffic.py); null when no matching profile is committed
        traffic, traffic_note = load_traffic(args.config, world, spp_step, sif, kernel_name)
        occ = r.kernel_occupancy()
        roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_note": traffic_note,
                    "kernel": kernel_name, "node_bytes": node_bytes,
                    # static wavefront occupancy of that kernel against the gfx950 limit (32 waves per CU)
                    "occupancy": {"waves_per_cu": int(occ.waves_per_cu), "max_waves_per_cu": int(occ.max_waves_per_cu),
                                  "frac": round(occ.waves_per_cu / max(1, occ.max_waves_per_cu), 3), "vgprs": int(occ.vgprs),
                                  "lds_bytes_per_block": int(occ.lds_bytes_per_block),
                                  "resident_blocks": int(occ.resident_grid_blocks), "compute_units": int(occ.compute_units)},
                    "avg_launch_ms": round(avg_ms, 4), "launches": int(st.intersect_launches),
                    "alg_bytes_per_launch": int(bytes_per_launch),
                    "rays_walked_frac": round(rays_walked / max(1, rays_sample), 3),
                    "node_visits_per_walked_ray": round(node_visits_sample / max(1, rays_walked), 2),
                    "tri_tests_per_walked_ray": round(tri_tests_sample / max(1, rays_walked), 2),
                    # divergence (SURVEY 8d's secondary limits): useful lane slots / issued lane slots of the node loop
                    # and of the cooperative triangle tests, from the instrumented instance on a batch of the first (up to 64) samples
                    "active_lane_frac": {"node_steps": round(trav.bvh_node_visits / max(1, trav.node_lane_slots), 3),
                                         "triangle_tests": round(trav.bvh_tri_tests / max(1, trav.tri_lane_slots), 3)},
                    "stage_ms": {"raygen": round(st.raygen_ms, 3), "intersect": round(st.intersect_ms, 3),
                                 "shade": round(st.shade_ms, 3), "accumulate": round(st.accumulate_ms, 3)}}

    # ---- CPU baseline: the oracle, timed on this box's host cores on a bounded sample (rank 0, N = 1) ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        # the GPU box gives one GPU slot a share of 16 host cores; PRT_CPU_THREADS overrides
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        cores = int(os.environ.get("PRT_CPU_THREADS", min(avail, 16)))
        osc = orc.OracleScene(scene.desc())
        cd = cam.desc()
        # calibrate on a thin strip, then size the sample (rows of the frame, or whole frames) for ~cpu_seconds
        y0 = H // 2
        tc = time.perf_counter()
        _, _, rays_c = osc.render(cd, W, H, spp=1, max_depth=max_depth, seed=0, iterative=False, use_bvh=True,
                                  n_threads=cores, rect=(0, y0 - 8, W, y0 + 8))
        tcal = max(time.perf_counter() - tc, 1e-6)
        frame_s = tcal * H / 16.0  # estimated seconds for one full-frame sample
        if frame_s > args.cpu_seconds:
            rows, spp_c = max(16, int(H * args.cpu_seconds / frame_s)), 1
        else:
            rows, spp_c = H, max(1, min(64, int(args.cpu_seconds / frame_s)))
        ya = max(0, (H - rows) // 2)
        tc = time.perf_counter()
        _, _, rays_b = osc.render(cd, W, H, spp=spp_c, max_depth=max_depth, seed=0, iterative=False, use_bvh=True,
                                  n_threads=cores, rect=(0, ya, W, ya + rows))
        tb = time.perf_counter() - tc
        cpu = {"value": round(rays_b / tb / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
               "sample": f"{args.config} rows {ya}..{ya + rows} of {H} at {spp_c} spp ({rays_b} rays, {tb:.1f} s, "
                         f"{cores} threads); oracle = CPU restatement of the reference CPU backend (recursive TraceRay) "
                         "using the oracle's own median-split BVH for the mesh; the reference itself has no BVH "
                         "(linear scan over all primitives) and is unbuildable here"}

    final = gather() if args.dump else None  # a collective: every rank takes part
    if args.dump and rank == 0:
        rgb, wts = final
        torch.cuda.synchronize()
        a = rgb.cpu().numpy().reshape(H, W, 3)
        w = wts.cpu().numpy().reshape(H, W)
        mean = np.where(w[..., None] > 0, a / np.maximum(w[..., None], 1e-30), 0).astype(np.float32)
        prt.write_pfm(args.dump + ".pfm", mean)
        rgba = torch.empty(H * W * 4, dtype=torch.uint8, device=device)
        r.film_tonemap(rgb.data_ptr(), wts.data_ptr(), rgba.data_ptr())
        r.synchronize()
        prt.write_ppm(args.dump + ".ppm", rgba.cpu().numpy().reshape(H, W, 4))

    if rank == 0:
        out = {
            "metric": "Mrays/sec at 1920x1080, 4 bounces, ~1M-tri scene",
            "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {MESH_OF[args.config]} = {n_tris} triangles + ground quad + emissive quad, "
                                   f"{W}x{H}, max_depth {max_depth} segments (= {max_depth - 1} bounces), "
                                   f"{spp_step} spp per step ({args.steps * spp_step} spp timed of the config's {spp_total}), "
                                   f"image tiled over {world} GPU(s) + per-step gather to rank 0",
                       "triangles": n_tris, "bvh_nodes": int(bvh.n_nodes), "bvh_max_depth": int(bvh.max_depth),
                       "bvh8_nodes": int(bvh.n_nodes8), "bvh8_depth": int(bvh.depth8),
                       "width": W, "height": H, "max_depth": max_depth, "spp_per_step": spp_step,
                       "samples_in_flight": sif, "seed": 0, "rays_timed": rays_total,
                       "rays_per_sample": rays_sample, "setup_s": round(setup_s, 2), "variant": args.variant},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
