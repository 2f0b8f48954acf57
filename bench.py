#!/usr/bin/env python3
"""bench.py — headline benchmark: Mrays/s of the wavefront path tracer on BASELINE.json's metric config.

  python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher: starts its N ranks as child processes)
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

The JSON line also carries:
  roofline      the dominant kernel (k_traverse8_persistent) against the roof that binds it: VECTOR-INSTRUCTION ISSUE.
                achieved = wave-level VALU instructions the algorithm needs per launch (node-loop and triangle-loop trip
                counts MEASURED by the instrumented instance on this input x the loops' static VALU counts from the
                kernel's ISA, tools/isa_count.py) / the launch time measured here with HIP events; peak = 1024 SIMDs x
                2.4 GHz / 2 cycles per wave64 instruction.  `valu_insts_measured` is the hardware's own count
                (SQ_INSTS_VALU per launch, committed rocprofv3 pass) and `hbm` the memory side (algorithmic bytes,
                counter-side bytes, fraction of the 8 TB/s HBM peak).
  secondary     the same pipeline on the other BASELINE configs that fit one GPU and on C3 with jittered primary rays,
                timed in this run at reduced sample counts (N = 1 only).
  cpu_baseline  the oracle on this box's host cores, (ii) the workload itself with the oracle's own BVH, and
                `reference_semantics`: (i) the reference's linear scan (primitive.cpp:26-49) on its default scene.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
# Vector-instruction issue roof: 256 CUs x 4 SIMD-32; a wave64 VALU instruction occupies its SIMD for 2 cycles
# (MI355X_MICROARCH.md "v_fma_f32 (wave64): 2 cyc"; = the 157.3 TFLOP/s fp32 vector peak / 128 flops per wave FMA)
N_SIMDS = 1024
CLOCK_GHZ = 2.4
VALU_PEAK_GINST = N_SIMDS * CLOCK_GHZ / 2.0  # 1228.8 G wave-instructions / s
NODE_BYTES = 80        # one compressed 8-wide node visit: origin + exponents + 8 x (6 quantized planes + meta) (csrc/bvh.h)
NODE_BYTES4 = 128      # one BVH4 node visit (A/B kernels): four child AABBs + four child refs = one cache line
TRI_BYTES = 48         # one leaf triangle test: 3 x float4 (P0+prim, P1+material, P2)
RAY_FIXED_BYTES = 44   # traversal kernel per ray it walks: origin+dir (2 x 16 B) + hit id and d2 read (8 B) + hit id write (4 B)
# static VALU instructions per trip of the traversal kernel's two loops when profiles/*_isa_counts.json is missing
# (tools/isa_count.py on the round-2 kernel: node step 258, triangle round 140)
VALU_FALLBACK = {"node_step": 258, "triangle_round": 140}

MESH_OF = {"C2": "bunny.ply refined by longest-edge bisection", "C3": "dragon.ply refined by longest-edge bisection",
           "C4": "dragon.ply refined by longest-edge bisection", "C5": "12 baked copies of the refined dragon.ply",
           "C5I": "12 placed copies (PrtInstance, two-level BVH) of the refined dragon.ply"}


def newest_profile(pattern, match):
    """Latest committed profiles/<pattern> whose keys equal `match` (dict); (data, filename) or (None, reason)."""
    best = (None, "no committed profile matches this configuration")
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern))):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        if all(t.get(k) == v for k, v in match.items()):
            best = (t, os.path.basename(f))
    return best


def kernel_sha():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from kernel_sha import kernel_sha as f
        return f()
    except Exception:
        return None


def isa_counts(instance):
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_isa_counts.json")), reverse=True):
        try:
            j = json.load(open(f))
            e = j["instances"][instance]
            return {"node_step": int(e["node_step"]["valu"]), "triangle_round": int(e["triangle_round"]["valu"]),
                    "source": os.path.basename(f), "kernel_sha16": j.get("kernel_sha16")}
        except Exception:
            continue
    return dict(VALU_FALLBACK, source="bench.py VALU_FALLBACK", kernel_sha16=None)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C3", help="C2 | C3 | C4 | C5 | C5I (SURVEY.md §8d; C5I = C5 as placed copies of one mesh)")
    ap.add_argument("--spp-per-step", type=int, default=0, help="0 = the config's full sample count (C3: 256)")
    ap.add_argument("--samples-in-flight", type=int, default=0, help="0 = auto")
    ap.add_argument("--jitter", type=int, default=0, help="1 = jittered primary rays (PrtSampling.jitter), an A/B / profiling option")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--wide", type=int, default=2, help="2 = compressed 8-wide tree (default), 1 = 4-wide tree (A/B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary block (other configs, jittered C3, presets)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="per-frame gather on the render stream instead of a side stream")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL over xGMI) | gloo (rehearsal on a box with fewer GPUs)")
    ap.add_argument("--dump", default="", help="write the final frame as PPM/PFM with this path prefix")
    return ap.parse_args()


def time_config(prt, torch, name, device, spp_step, steps, sampling=None, params=None):
    """One secondary measurement on this GPU: `steps` timed steps of `spp_step` samples of config `name`."""
    t0 = time.time()
    scene, cam, W, H, spp_total, max_depth = prt.scenes.config(name)
    film = prt.Film(W, H)
    r = prt.HipWavefrontRenderer(device=device, max_depth=max_depth, seed=0)
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    for k, v in (params or {}).items():
        r.set_param(k, v)
    r.Init(film, scene, cam)
    if sampling:
        r.set_sampling(**sampling)
    r.set_samples_in_flight(spp_step)
    setup = time.time() - t0
    r.render_async(spp_step)
    torch.cuda.synchronize()
    r.synchronize()  # raises on a tripped traversal watchdog / full overflow list
    r.reset_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        r.render_async(spp_step)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    r.synchronize()
    st = r.stats()
    out = {"value": round(int(st.rays_total) / dt / 1e6, 1), "unit": "Mrays/s", "ms_per_step": round(dt / steps * 1e3, 3),
           "steps": steps, "spp_per_step": spp_step, "rays_timed": int(st.rays_total), "width": W, "height": H,
           "max_depth": max_depth, "triangles": scene.n_triangles, "setup_s": round(setup, 2)}
    if sampling:
        out["sampling"] = sampling
    film._renderer = None
    del r, film, scene
    import gc
    gc.collect()
    return out


def presets_block(prt, torch, orc, device, cores, cpu_seconds):
    """SURVEY §8d CPU-baseline flavour (i): the reference's own default scene (RANDOM_BALLS_LARGE, src/core/scene.h:20:
    809 analytic primitives) at 1920x1080 with the CPU backend's 20 segments: the reference-semantics linear scan
    (primitive.cpp:26-49) on the host cores next to the GPU with the same linear scan (prim_bvh = 0) and with the BVH over
    the primitives (default)."""
    W, H, depth = 1920, 1080, 20
    scene = prt.Scene("RANDOM_BALLS_LARGE")
    cam = prt.Camera(width=W, height=H)
    out = {"scene": "RANDOM_BALLS_LARGE (809 analytic primitives), 1920x1080, max_depth 20, camera of src/main.cpp:142-150"}
    for key, pb, spp in (("gpu_linear_scan", 0, 4), ("gpu_primitive_bvh", 1, 16)):
        film = prt.Film(W, H)
        r = prt.HipWavefrontRenderer(device=device, max_depth=depth, seed=0)
        r.set_stream(torch.cuda.current_stream().cuda_stream)
        r.set_param("prim_bvh", pb)
        r.Init(film, scene, cam)
        r.set_samples_in_flight(spp)
        r.render_async(spp)
        torch.cuda.synchronize()
        r.synchronize()
        r.reset_stats()
        t0 = time.perf_counter()
        r.render_async(spp)
        r.render_async(spp)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        r.synchronize()
        out[key] = {"value": round(int(r.stats().rays_total) / dt / 1e6, 1), "unit": "Mrays/s", "spp_timed": 2 * spp}
        film._renderer = None
        del r, film
    if orc is not None:
        osc = orc.OracleScene(scene.desc())
        y0 = H // 2
        tc = time.perf_counter()
        osc.render(cam.desc(), W, H, spp=1, max_depth=depth, seed=0, iterative=False, use_bvh=False, n_threads=cores,
                   rect=(0, y0 - 4, W, y0 + 4))
        tcal = max(time.perf_counter() - tc, 1e-6)
        rows = int(max(8, min(H, 8 * cpu_seconds / tcal)))
        ya = max(0, (H - rows) // 2)
        tc = time.perf_counter()
        _, _, rays = osc.render(cam.desc(), W, H, spp=1, max_depth=depth, seed=0, iterative=False, use_bvh=False,
                                n_threads=cores, rect=(0, ya, W, ya + rows))
        tb = time.perf_counter() - tc
        out["cpu_linear_scan"] = {"value": round(rays / tb / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
                                  "sample": f"rows {ya}..{ya + rows} of {H} at 1 spp ({rays} rays, {tb:.1f} s, {cores} threads); "
                                            "recursive TraceRay + PrimitiveList::Intersect's linear scan, as the reference's CPU backend"}
    return out


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU) as CHILD processes of this
    one and forward rank 0's JSON line.  Nothing in this parent touches the GPU (no torch import, no HIP call: the build is
    a hipcc / g++ child process), so no process that has initialised the GPU is ever replaced or forked.  Any rank
    failing fails the run: the others are terminated and the exit code is non-zero."""
    import socket
    import subprocess

    import __graft_entry__ as g
    g.build(quiet=True, load=False)  # once, here: the ranks then find everything up to date
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for rk in range(args.gpus):
        env = dict(os.environ, RANK=str(rk), LOCAL_RANK=str(rk), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PRT_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if rk == 0 else subprocess.DEVNULL, text=True))
    out0 = ""
    rc = 0
    try:
        # rank 0 prints the line after the last collective; poll all ranks so that one dying early ends the run instead
        # of leaving the others in a rendezvous
        live = set(range(args.gpus))
        while live:
            for rk in sorted(live):
                p = procs[rk]
                try:
                    if rk == 0:
                        o, _ = p.communicate(timeout=0.5)
                        out0 += o or ""
                    else:
                        p.wait(timeout=0.5)
                except subprocess.TimeoutExpired:
                    continue
                live.discard(rk)
                if p.returncode != 0:
                    rc = rc or p.returncode or 1
                    print(f"bench.py: rank {rk} exited with code {p.returncode}", file=sys.stderr)
                    for q in procs:
                        if q.poll() is None:
                            q.terminate()
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    sys.stdout.write(out0)
    sys.stdout.flush()
    if rc == 0 and not any(l.startswith("{") for l in out0.splitlines()):
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        rc = 1
    raise SystemExit(rc)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)  # does not return
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
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE = {world}: launch N ranks with torch.distributed.run "
                         "--nproc-per-node N for --gpus N (one process per GPU)")
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
    rdev = device if (world > 1 and args.backend == "nccl") else "cpu"
    ranks_seen = 1
    if world > 1:  # how many ranks the collective layer really connects (the driver checks it against --gpus)
        one = torch.ones(1, dtype=torch.int64, device=rdev)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        ranks_seen = int(one.item())

    # ---- workload (outside the timed region: PLY parse, refinement, BVH build, upload) ----
    t_setup = time.time()
    scene, cam, W, H, spp_total, max_depth = prt.scenes.config(args.config)
    film = prt.Film(W, H)
    r = prt.HipWavefrontRenderer(device=local_rank, max_depth=max_depth, seed=0, rank=rank, world_size=world)
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    r.Init(film, scene, cam)
    r.set_variant(args.variant)
    r.set_param("wide", args.wide)
    if args.jitter:
        r.set_sampling(jitter=1)
    if os.environ.get("PRT_FUSE"):  # A/B: 0 = the producers store every ray (no fused analytic segment)
        r.set_param("fuse", int(os.environ["PRT_FUSE"]))
    for kv in filter(None, os.environ.get("PRT_PARAMS", "").split(",")):  # A/B: PRT_PARAMS=name=value,...
        k, v = kv.split("=")
        r.set_param(k, int(v))
    n_tris = scene.n_triangles
    bvh = r.bvh_info()
    spp_step = args.spp_per_step or spp_total  # a step = one complete frame of the config
    # many samples in flight: the late bounces are latency/tail-bound, more rays per launch hide it (measured on C3 with
    # the current kernels: 64 -> 13.5, 128 -> 14.2, 256 -> 14.3 Grays/s); 288 GB of HBM make 68 GB of path state cheap
    sif = args.samples_in_flight or min(spp_step, max(1, (256 * 1920 * 1080 * world) // (W * H)))
    sif = max(1, min(sif, spp_step))
    r.set_samples_in_flight(sif)
    gather = prt.dist.FilmGather(r, device, overlap=not args.no_overlap)
    setup_s = time.time() - t_setup

    # trip counts and algorithmic traffic of the dominant kernel, measured with the instrumented traversal on the first
    # samples (a batch of up to 64 samples, like the timed batches: with compact primary rays a wave of bounce 0 holds 64
    # samples of one pixel, which a one-sample batch cannot show in the active-lane fractions; counts are per sample below)
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
    node_wave_steps_sample = int(trav.node_lane_slots) / 64.0 / n_meas   # trips of the node loop, summed over waves
    tri_rounds_sample = int(trav.tri_lane_slots) / 64.0 / n_meas         # trips of the triangle loop

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
    r.synchronize()  # a tripped traversal watchdog or a full overflow list is an error here, not a fast-looking number
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
    r.synchronize()  # outside the timed region: reads the watchdog / overflow flag of the timed launches
    st = r.stats()
    rays_local = int(st.rays_total)

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        c = torch.tensor([rays_local], dtype=torch.int64, device=rdev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        rays_total = int(c.item())
    else:
        rays_total = rays_local

    value = rays_total / dt / 1e6

    # ---- roofline of the dominant kernel, this rank ----
    roofline = None
    if not args.no_kernel_timing and st.intersect_launches:
        avg_ms = st.intersect_ms / st.intersect_launches
        # measure_traversal ran on THIS rank's tiles, so the per-sample figures are already the local share; a launch is one
        # depth of one batch of `sif` samples: per launch = per sample x samples / launches
        per_launch = st.samples / st.intersect_launches
        inst = r.kernel_instance()  # what prt_launch_traverse launches for this scene (one decision function, csrc/prt_kernels.hip)
        isa = isa_counts(inst)
        valu_alg = (node_wave_steps_sample * isa["node_step"] + tri_rounds_sample * isa["triangle_round"]) * per_launch
        achieved = valu_alg / (avg_ms * 1e-3) / 1e9
        bytes_per_launch = alg_bytes_sample * per_launch
        match = {"config": args.config, "n_gpus": world, "samples_in_flight": sif, "kernel": kernel_name, "jitter": args.jitter}
        # counter side, from separate rocprofv3 --pmc passes of this same command (tools/profile_round.sh -> profiles/)
        traffic_p, traffic_src = newest_profile("*_traffic.json", match)
        sq_p, sq_src = newest_profile("*_sq.json", match)
        traffic = traffic_p.get("hbm_bytes_per_launch") if traffic_p else None
        occ = r.kernel_occupancy()
        # the roof the kernel is closest to: vector-instruction issue on the cache-resident configs (C2-C4), the fabric /
        # HBM side once the tree outgrows the 256 MB Infinity Cache (C5); both fractions are always in the line
        hbm_frac = (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None
        valu_frac = achieved / VALU_PEAK_GINST
        if hbm_frac is not None and hbm_frac > valu_frac:
            head = {"bound": "hbm", "achieved": round(traffic / (avg_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(hbm_frac, 4)}
        else:
            head = {"bound": "valu", "achieved": round(achieved, 2), "peak": VALU_PEAK_GINST, "unit": "G wave-instr/s",
                    "frac": round(valu_frac, 4)}
        # evidence from committed profiles is only as good as the kernel code it was taken on: every file carries the
        # fingerprint of the kernel sources (tools/kernel_sha.py); anything that does not match what runs here is named
        sha_now = kernel_sha()
        stale = [n for n, p_ in (("traffic", traffic_p), ("sq", sq_p), ("isa_counts", isa)) if p_ and p_.get("kernel_sha16") != sha_now]
        roofline = {
            **head,
            "kernel_sha16": sha_now, "stale": bool(stale), "stale_sources": stale,
            "traffic": traffic, "traffic_note": (traffic_src + ": " + traffic_p.get("note", "")) if traffic_p else traffic_src,
            "kernel": kernel_name, "kernel_instance": inst,
            "avg_launch_ms": round(avg_ms, 4), "launches": int(st.intersect_launches),
            "valu": {"achieved_Ginst_s": round(achieved, 2), "peak_Ginst_s": VALU_PEAK_GINST, "frac": round(valu_frac, 4),
                     "alg_insts_per_launch": int(valu_alg),
                     "node_wave_steps_per_launch": int(node_wave_steps_sample * per_launch),
                     "triangle_rounds_per_launch": int(tri_rounds_sample * per_launch),
                     "valu_per_node_step": isa["node_step"], "valu_per_triangle_round": isa["triangle_round"],
                     "isa_source": isa["source"],
                     "peak_def": "1024 SIMD-32 x 2.4 GHz / 2 cycles per wave64 VALU instruction",
                     # the same achieved rate against the roof the round-1 review wrote down (a wave64 instruction every 4
                     # cycles per SIMD: 614.4 G/s); the line's own frac uses the stricter 2-cycle figure, which is what the
                     # fp32 vector peak of the part implies and what v_fma_f32 was measured to sustain (profiles/r2_issue_rate.txt)
                     "frac_at_4_cycles_per_instr": round(valu_frac * 2.0, 4),
                     # the hardware's own count of the same launches, and where the wave cycles went (committed PMC pass)
                     "valu_insts_measured": sq_p.get("valu_insts_per_launch") if sq_p else None,
                     "frac_measured": round(sq_p["valu_insts_per_launch"] / (avg_ms * 1e-3) / 1e9 / VALU_PEAK_GINST, 4) if sq_p else None,
                     "sq": {k: sq_p[k] for k in ("wave_cycles_share", "mean_occupancy_per_cu", "l2_hit_rate") if k in sq_p} if sq_p else None,
                     "sq_source": sq_src},
            "hbm": {"alg_bytes_per_launch": int(bytes_per_launch), "alg_GBs": round(bytes_per_launch / (avg_ms * 1e-3) / 1e9, 1),
                    "traffic_bytes_per_launch": traffic,
                    "hbm_frac": round(hbm_frac, 4) if hbm_frac is not None else None,
                    "peak_GBs": HBM_PEAK_GBS, "node_bytes": node_bytes,
                    "note": "alg = 80 B x node visits + 48 B x triangle tests + 44 B x walked rays; served mostly by L2 / Infinity "
                            "Cache on C2-C4 (so alg_GBs may exceed the HBM peak); hbm_frac = fabric-side counter bytes / time / peak"},
            # the CU's vector-memory path: per-lane 16-B gathers (5 per node visit, 3 per triangle test) per microsecond per CU,
            # beside what a loop that only gathers reaches on this chip (tools/gather_rate.hip, profiles/r2_gather_rate.txt:
            # measured capacities, no datasheet figure; TUNING.md "What bounds bounces >= 1").  Averaged over all launches of the
            # step, i.e. including bounce 0, whose 64 lanes share one line per instruction
            "vmem": {"lane_loads_per_launch": int((5 * node_visits_sample + 3 * tri_tests_sample) * per_launch),
                     "lane_loads_per_us_per_cu": round((5 * node_visits_sample + 3 * tri_tests_sample) * per_launch
                                                       / (avg_ms * 1e3) / max(1, int(occ.compute_units)), 1),
                     "gather_only_loop_per_us_per_cu": {"l1": 8900, "l2": 3700, "16MiB_table": 2600, "infinity_cache": 2400, "hbm_1GiB_table": 1540},
                     "source": "profiles/r2_gather_rate.txt"},
            # static wavefront occupancy of that kernel against the gfx950 limit (32 waves per CU)
            "occupancy": {"waves_per_cu": int(occ.waves_per_cu), "max_waves_per_cu": int(occ.max_waves_per_cu),
                          "frac": round(occ.waves_per_cu / max(1, occ.max_waves_per_cu), 3), "vgprs": int(occ.vgprs),
                          "lds_bytes_per_block": int(occ.lds_bytes_per_block),
                          "resident_blocks": int(occ.resident_grid_blocks), "compute_units": int(occ.compute_units)},
            "rays_walked_frac": round(rays_walked / max(1, rays_sample), 3),
            "walked_rays_rate_Mrays_s": round(value * rays_walked / max(1, rays_sample), 1),
            "node_visits_per_walked_ray": round(node_visits_sample / max(1, rays_walked), 2),
            "tri_tests_per_walked_ray": round(tri_tests_sample / max(1, rays_walked), 2),
            # divergence (SURVEY 8d's secondary limits): useful lane slots / issued lane slots of the node loop
            # and of the cooperative triangle tests, from the instrumented instance on a batch of the first (up to 64) samples
            "active_lane_frac": {"node_steps": round(trav.bvh_node_visits / max(1, trav.node_lane_slots), 3),
                                 "triangle_tests": round(trav.bvh_tri_tests / max(1, trav.tri_lane_slots), 3)},
            "stage_ms": {"raygen": round(st.raygen_ms, 3), "intersect": round(st.intersect_ms, 3),
                         "shade": round(st.shade_ms, 3), "accumulate": round(st.accumulate_ms, 3)}}

    final = None
    if args.dump:  # a collective: every rank takes part
        final = gather()
        gather.wait()
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
    scene_desc, cam_desc = scene.desc(), cam.desc()
    film._renderer = None  # (Film and renderer refer to each other)
    del gather, r, film    # the headline run's 68 GB of path state go before the secondary runs allocate theirs
    import gc
    gc.collect()

    # ---- CPU baseline: the oracle, timed on this box's host cores on a bounded sample (rank 0, N = 1) ----
    cpu = None
    orc = None
    cores = 1
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        # the GPU box gives one GPU slot a share of 16 host cores; PRT_CPU_THREADS overrides
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        cores = int(os.environ.get("PRT_CPU_THREADS", min(avail, 16)))
        osc = orc.OracleScene(scene_desc)
        # calibrate on a thin strip, then size the sample (rows of the frame, or whole frames) for ~cpu_seconds
        y0 = H // 2
        tc = time.perf_counter()
        _, _, rays_c = osc.render(cam_desc, W, H, spp=1, max_depth=max_depth, seed=0, iterative=False, use_bvh=True,
                                  n_threads=cores, rect=(0, y0 - 8, W, y0 + 8))
        tcal = max(time.perf_counter() - tc, 1e-6)
        frame_s = tcal * H / 16.0  # estimated seconds for one full-frame sample
        if frame_s > args.cpu_seconds:
            rows, spp_c = max(16, int(H * args.cpu_seconds / frame_s)), 1
        else:
            rows, spp_c = H, max(1, min(64, int(args.cpu_seconds / frame_s)))
        ya = max(0, (H - rows) // 2)
        tc = time.perf_counter()
        _, _, rays_b = osc.render(cam_desc, W, H, spp=spp_c, max_depth=max_depth, seed=0, iterative=False, use_bvh=True,
                                  n_threads=cores, rect=(0, ya, W, ya + rows))
        tb = time.perf_counter() - tc
        cpu = {"value": round(rays_b / tb / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
               "sample": f"{args.config} rows {ya}..{ya + rows} of {H} at {spp_c} spp ({rays_b} rays, {tb:.1f} s, "
                         f"{cores} threads); oracle = CPU restatement of the reference CPU backend (recursive TraceRay) "
                         "using the oracle's own median-split BVH for the mesh (not the GPU's tree: the oracle may not "
                         "touch product code); the reference itself has no BVH (linear scan over all primitives) and is "
                         "unbuildable here"}
        del osc

    # ---- secondary block (N = 1): the other single-GPU configs, jittered C3, the reference's default scene ----
    secondary = None
    if rank == 0 and world == 1 and not args.no_secondary:
        secondary = {"note": "same pipeline, timed in this run with inputs resident; reduced sample counts (spp_per_step x steps; "
                             "C3_jitter: the headline's own 256 samples per step), spp_per_step samples in flight; the headline "
                             "stays C3 without jitter"}
        secondary["C3_jitter"] = time_config(prt, torch, "C3", local_rank, 256, 3, sampling={"jitter": 1})
        secondary["C2"] = time_config(prt, torch, "C2", local_rank, 64, 3)
        secondary["C5"] = time_config(prt, torch, "C5", local_rank, 64, 2)
        secondary["C5I"] = time_config(prt, torch, "C5I", local_rank, 64, 2)
        # BASELINE config 4 (the C3 scene at 3840x2160, 8 bounces; the config's 8 GPUs tile the frame, here the whole frame on one)
        secondary["C4"] = time_config(prt, torch, "C4", local_rank, 16, 2)
        pre = presets_block(prt, torch, orc, local_rank, cores, min(5.0, args.cpu_seconds))
        secondary["RANDOM_BALLS_LARGE"] = {k: v for k, v in pre.items() if k != "cpu_linear_scan"}
        if cpu is not None and "cpu_linear_scan" in pre:
            cpu["reference_semantics"] = dict(pre["cpu_linear_scan"], scene=pre["scene"],
                                              gpu_same_scan_Mrays_s=pre["gpu_linear_scan"]["value"],
                                              gpu_primitive_bvh_Mrays_s=pre["gpu_primitive_bvh"]["value"])

    if rank == 0:
        out = {
            "metric": "Mrays/sec at 1920x1080, 4 bounces, ~1M-tri scene",
            "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "ranks_seen": ranks_seen,
            "config": {"workload": f"{args.config}: {MESH_OF[args.config]} = {n_tris} triangles + ground quad + emissive quad, "
                                   f"{W}x{H}, max_depth {max_depth} segments (= {max_depth - 1} bounces), "
                                   f"{spp_step} spp per step ({args.steps * spp_step} spp timed of the config's {spp_total}), "
                                   f"image tiled over {world} GPU(s) + per-step gather to rank 0"
                                   + (", jittered primary rays" if args.jitter else ""),
                       "triangles": n_tris, "bvh_nodes": int(bvh.n_nodes), "bvh_max_depth": int(bvh.max_depth),
                       "bvh8_nodes": int(bvh.n_nodes8), "bvh8_depth": int(bvh.depth8),
                       "width": W, "height": H, "max_depth": max_depth, "spp_per_step": spp_step,
                       "samples_in_flight": sif, "seed": 0, "jitter": args.jitter, "rays_timed": rays_total,
                       "rays_per_sample": rays_sample, "setup_s": round(setup_s, 2), "variant": args.variant,
                       "gather": "side stream from a snapshot" if (world > 1 and not args.no_overlap and args.backend == "nccl") else "render stream"},
            "roofline": roofline, "cpu_baseline": cpu, "secondary": secondary,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
