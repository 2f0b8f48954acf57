"""N>1 path on CPU: two gloo ranks tile the image, gather the per-tile payloads to rank 0 and un-tile them.
The pixel values come from the oracle (the checker) so the assembled image can be compared with the full-frame
oracle render; the tile layout code is the same module bench.py uses on the GPUs (parallelraytracing_amd/dist.py)."""
import os
import subprocess
import sys


import util

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
import util
from util import orc, prt
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
W, H, spp, depth = 52, 36, 2, 4
scene = prt.Scene("DEFAULT")
osc = util.oracle_scene(scene)
cam = prt.Camera(width=W, height=H).desc()
# every rank renders ONLY the pixels of its own 8x8 tiles (the GPU ranks do the same with k_raygen's tile map)
owner, slot = prt.dist.pixel_owner_and_slot(W, H, world)
acc = np.zeros((H, W, 3), np.float32); wts = np.zeros((H, W), np.float32)
tx, ty, stride = prt.dist.tile_layout(W, H, world)
rays_local = 0
for t in range(rank, tx * ty, world):
    x0, y0 = (t % tx) * 8, (t // tx) * 8
    _, _, rays = osc.render(cam, W, H, spp=spp, max_depth=depth, seed=3, iterative=True,
                            rect=(x0, y0, min(W, x0 + 8), min(H, y0 + 8)), accum=acc, weights=wts)
    rays_local += rays
assert (wts[owner != rank] == 0).all() and (wts[owner == rank] == spp).all()
payload = torch.from_numpy(prt.dist.pack_tiles_numpy(acc, wts, rank, world))
parts = [torch.empty_like(payload) for _ in range(world)] if rank == 0 else None
dist.gather(payload, parts, dst=0)                      # the one collective of the path
cnt = torch.tensor([rays_local], dtype=torch.int64)
dist.all_reduce(cnt)                                    # ray counters
if rank == 0:
    g = torch.stack(parts).numpy()
    a2, w2 = prt.dist.untile_numpy(g, W, H)
    full, wfull, rays_full = osc.render(cam, W, H, spp=spp, max_depth=depth, seed=3, iterative=True)
    assert np.array_equal(a2, full) and np.array_equal(w2, wfull), "assembled image differs from the single-rank image"
    assert int(cnt.item()) == rays_full
    print("GLOO_OK", int(cnt.item()))
dist.destroy_process_group()
'''


def _run(world):
    port = 29500 + (os.getpid() % 1000)
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER, util.ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    return outs


def test_two_rank_tile_gather_on_gloo():
    outs = _run(2)
    assert "GLOO_OK" in outs[0]


def test_three_rank_tile_gather_on_gloo():
    outs = _run(3)  # tiles do not divide evenly: exercises the padded payload stride
    assert "GLOO_OK" in outs[0]


def test_bench_self_launch_forwards_a_rank_failure_as_a_non_zero_exit():
    """`python bench.py --gpus 2` with no launcher starts its ranks as child processes (the parent touches no GPU).  Here,
    without a GPU, every rank stops with "needs an MI355X": the parent must report it and exit non-zero, not hang in a
    rendezvous and not print a result line.  (On the GPU box tests/test_gpu_dist.py runs the same entry to a real frame.)"""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("the GPU box runs the successful case instead (tests/test_gpu_dist.py)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--backend", "gloo", "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0
    assert "exited with code" in p.stderr and "needs an MI355X" in p.stderr
    assert not any(l.startswith("{") for l in p.stdout.splitlines())
