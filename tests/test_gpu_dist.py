"""The RCCL side of the N > 1 path on the one GPU a test box has (-m gpu).

A whole 8-GPU node is never available to these tests (two RCCL ranks cannot share a device), so what CAN run does:
one rank on the "nccl" backend doing everything bench.py's multi-GPU step does (worker script), and the 2-rank gather
rehearsed over gloo with both ranks rendering on the same GPU (frame bit-identical to the 1-rank frame).  The reference
has no multi-GPU path at all (cudaSetDevice(0), src/backend/optix/renderer.cpp:217)."""
import os
import socket
import subprocess
import sys

import pytest

import util

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_one_rank_on_the_rccl_backend_gathers_and_resolves():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(util.ROOT, "tests", "nccl_world1_worker.py"), str(_free_port())],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0 and "NCCL_WORLD1_OK" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])


def test_bench_two_ranks_sharing_the_gpu_over_gloo_matches_one_rank(tmp_path):
    """bench.py itself, launched the way the driver launches it for N = 2 (torch.distributed.run), both ranks on GPU 0
    with the gloo transport: the JSON line must carry ranks_seen = 2 and the 2-rank frame must equal the 1-rank frame."""
    import json

    import numpy as np
    outs = {}
    # "self": plain `python bench.py --gpus 2` with no launcher and no WORLD_SIZE: bench.py starts its two ranks itself
    for n in (1, 2, "self"):
        dump = str(tmp_path / f"f{n}")
        args = ["--gpus", "1" if n == 1 else "2", "--steps", "1", "--warmup", "0", "--config", "C2", "--spp-per-step", "2",
                "--no-cpu-baseline", "--no-secondary", "--dump", dump]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PRT_BENCH_SAME_DEVICE="1")
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        if n == 1:
            cmd = [sys.executable, os.path.join(util.ROOT, "bench.py")] + args
        elif n == "self":
            cmd = [sys.executable, os.path.join(util.ROOT, "bench.py")] + args + ["--backend", "gloo"]
        else:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                   "--master-port", str(_free_port()), os.path.join(util.ROOT, "bench.py")] + args + ["--backend", "gloo"]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
        line = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')][-1]
        outs[n] = (json.loads(line), open(dump + ".pfm", "rb").read())
    assert outs[1][0]["ranks_seen"] == 1
    roof = outs[1][0]["roofline"]  # the line carries the three sides of the dominant kernel
    assert 0.0 < roof["frac"] <= 1.0 and roof["hbm"]["alg_bytes_per_launch"] > 0
    assert roof["vmem"]["lane_loads_per_launch"] > 0 and roof["vmem"]["lane_loads_per_us_per_cu"] > 0
    a = np.frombuffer(outs[1][1][-1280 * 720 * 12:], "<f4")
    for n in (2, "self"):
        assert outs[n][0]["ranks_seen"] == 2 and outs[n][0]["n_gpus"] == 2, n
        assert outs[1][0]["config"]["rays_timed"] == outs[n][0]["config"]["rays_timed"], n
        assert np.array_equal(a, np.frombuffer(outs[n][1][-1280 * 720 * 12:], "<f4")), n


# ---- the C/C++ multi-GPU host path (include/prt.h prt_group_*) ----------------------------------------------------------
def _frame(renderer_factory, scene, cam, W, H, depth, spp, seed, sampling=None):
    import numpy as np
    import parallelraytracing_amd as prt
    film = prt.Film(W, H)
    r = renderer_factory()
    r.max_depth, r.seed = depth, seed
    r.Init(film, scene, cam)
    if sampling:
        r.set_sampling(**sampling)
    for _ in range(spp):
        r.ProgressiveRender()  # one sample per call: a gather per call in the group renderer
    r.download()
    return r, film, np.array(film.accum), np.array(film.weights)


@pytest.mark.parametrize("n", [2, 3, 8])
def test_group_of_n_contexts_on_one_gpu_equals_one_context(n):
    """prt_group_*: n contexts (one host thread each) tile the image, the scene is built once and cloned, the payloads are
    gathered to rank 0 by peer copies and un-tiled: bit-identical to the single-context renderer, and to the oracle; ray
    counts add up.  (n ranks sharing ONE device is the only multi-rank form a 1-GPU box can run: transport "peer".)"""
    import numpy as np
    import parallelraytracing_amd as prt
    W, H, depth, spp, seed = 203, 117, 6, 3, 5  # not multiples of the 8x8 tile; more tiles than ranks
    scene = prt.scenes.mesh_scene(prt.scenes.refined("bunny.ply", 20_000))
    cam = prt.Camera(position=(2.0, 1.5, 3.0), width=W, height=H)
    r1, f1, a1, w1 = _frame(lambda: prt.HipWavefrontRenderer(device=0), scene, cam, W, H, depth, spp, seed)
    rn, fn, an, wn = _frame(lambda: prt.HipWavefrontGroupRenderer([0] * n), scene, cam, W, H, depth, spp, seed)
    assert rn.transport == "peer" and rn.n_devices == n
    assert np.array_equal(an, a1) and np.array_equal(wn, w1) and (wn == spp).all()
    assert rn.stats().rays_total == r1.stats().rays_total
    acc, wts, rays = util.oracle_scene(scene).render(cam.desc(), W, H, spp=spp, max_depth=depth, seed=seed, iterative=True,
                                                     use_bvh=True, n_threads=8)
    assert np.array_equal(an, acc) and rn.stats().rays_total == rays
    disp = rn.UpdateDisplay()
    assert np.array_equal(disp, r1.UpdateDisplay())
    # jittered + roulette, batched: still independent of the rank count
    rn.Clear()
    r1.film.Clear()
    r1.frame_index = 0
    sp = dict(jitter=1, rr_depth=2, clamp=0.0)
    rn.set_sampling(**sp)
    r1.set_sampling(**sp)
    rn.set_samples_in_flight(4)
    rn.ProgressiveRender(4)
    r1.ProgressiveRender(4)
    assert np.array_equal(rn.download().accum, r1.download().accum)


def test_group_single_rank_through_rccl():
    """PRT_GROUP_TRANSPORT=rccl with one rank: librccl is loaded on demand, ncclCommInitAll builds a 1-rank communicator
    and the gather runs as ncclAllGather on the context's stream: the RCCL branch of the C host path on real hardware."""
    code = (
        "import os, sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import parallelraytracing_amd as prt\n"
        "W, H = 160, 96\n"
        "scene = prt.Scene('MATERIAL_TEST'); cam = prt.Camera(width=W, height=H)\n"
        "f1 = prt.Film(W, H); r1 = prt.HipWavefrontRenderer(device=0, max_depth=5, seed=2); r1.Init(f1, scene, cam); r1.ProgressiveRender(3); r1.download()\n"
        "fg = prt.Film(W, H); rg = prt.HipWavefrontGroupRenderer([0], max_depth=5, seed=2); rg.Init(fg, scene, cam)\n"
        "assert rg.transport == 'rccl', rg.transport\n"
        "rg.ProgressiveRender(3); rg.download()\n"
        "assert np.array_equal(fg.accum, f1.accum) and np.array_equal(fg.weights, f1.weights)\n"
        "print('GROUP_RCCL_OK')\n" % util.ROOT)
    env = dict(os.environ, PRT_GROUP_TRANSPORT="rccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0 and "GROUP_RCCL_OK" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])


def test_cpp_cli_tiles_the_frame_over_three_ranks(tmp_path):
    """The C++ adapter (host/prt_renderer.hpp HipWavefrontRenderer(devices)) through the CLI: --devices 0,0,0 must write
    the same PFM as --devices 0."""
    exe = os.path.join(util.ROOT, "parallelraytracing_amd", "csrc", "prt_render")
    outs = []
    for devs in ("0", "0,0,0"):
        out = str(tmp_path / ("f" + str(len(devs))))
        p = subprocess.run([exe, "--ply", os.path.join(util.ROOT, "assets", "models", "bunny.ply"), "--width", "200", "--height", "120",
                            "--spp", "3", "--depth", "5", "--seed", "4", "--devices", devs, "--out", out], capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        outs.append((open(out + ".pfm", "rb").read(), open(out + ".ppm", "rb").read(), p.stdout))
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1]
    assert "1 GPU(s) [gather: none]" in outs[0][2] and "3 GPU(s) [gather: peer]" in outs[1][2]
    assert outs[0][2].split("rays")[0].split(":")[-1] == outs[1][2].split("rays")[0].split(":")[-1]  # same ray count


def test_group_refit_on_two_contexts_matches_a_single_context():
    """prt_group_refit_meshes: both ranks refit their copy of the tree; the assembled frame of the deformed mesh equals the
    single-context frame (which tests/test_gpu_parity.py pins to the oracle)."""
    import numpy as np
    import parallelraytracing_amd as prt
    base = prt.scenes.refined("bunny.ply", 12_000)
    v = base.GetVertices()
    v[:, 0] += 0.02 * np.sin(3.0 * v[:, 1])
    moved = prt.Mesh(vertices=v, normals=base.GetNormals(), indices=base.GetIndices())
    W, H, depth, spp = 72, 48, 4, 2
    cam = prt.Camera(position=(2.0, 1.5, 3.0), width=W, height=H)
    frames = []
    for make in (lambda: prt.HipWavefrontRenderer(device=0, max_depth=depth, seed=5),
                 lambda: prt.HipWavefrontGroupRenderer([0, 0], max_depth=depth, seed=5)):
        film = prt.Film(W, H)
        r = make()
        r.max_depth, r.seed = depth, 5
        r.Init(film, prt.scenes.mesh_scene(base), cam)
        r.ProgressiveRender(1)
        r.Refit(prt.scenes.mesh_scene(moved))
        r.Clear() if hasattr(r, "Clear") else film.Clear()
        r.frame_index = 0
        for _ in range(spp):
            r.ProgressiveRender()
        r.download()
        frames.append(np.array(film.accum))
    assert np.array_equal(frames[0], frames[1]) and frames[0].any()
