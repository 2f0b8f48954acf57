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
    for n in (1, 2):
        dump = str(tmp_path / f"f{n}")
        args = ["--gpus", str(n), "--steps", "1", "--warmup", "0", "--config", "C2", "--spp-per-step", "2", "--no-cpu-baseline",
                "--no-secondary", "--dump", dump]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PRT_BENCH_SAME_DEVICE="1")
        if n == 1:
            cmd = [sys.executable, os.path.join(util.ROOT, "bench.py")] + args
        else:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                   "--master-port", str(_free_port()), os.path.join(util.ROOT, "bench.py")] + args + ["--backend", "gloo"]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
        line = [l for l in p.stdout.splitlines() if l.startswith('{"metric"')][-1]
        outs[n] = (json.loads(line), open(dump + ".pfm", "rb").read())
    assert outs[1][0]["ranks_seen"] == 1 and outs[2][0]["ranks_seen"] == 2 and outs[2][0]["n_gpus"] == 2
    assert outs[1][0]["config"]["rays_timed"] == outs[2][0]["config"]["rays_timed"]
    a = np.frombuffer(outs[1][1][-1280 * 720 * 12:], "<f4")
    b = np.frombuffer(outs[2][1][-1280 * 720 * 12:], "<f4")
    assert np.array_equal(a, b)
