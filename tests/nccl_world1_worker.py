"""Worker of test_gpu_dist.py: ONE rank on the RCCL ("nccl") backend.  Everything the N > 1 bench path does per frame
executes here on real hardware with world_size 1: init_process_group("nccl") bound to the device, the zero-copy view of
prt_film_local as a torch tensor, the snapshot + dist.gather on the side stream, prt_film_resolve_on, the all-reduces of
the ray counters, and the bit-for-bit comparison of the assembled frame with prt_film_read."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import parallelraytracing_amd as prt  # noqa: E402


def main():
    port = int(sys.argv[1])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    W, H, depth, spp = 200, 120, 5, 3
    scene = prt.Scene("MATERIAL_TEST")
    film = prt.Film(W, H)
    r = prt.HipWavefrontRenderer(device=0, max_depth=depth, seed=6, rank=0, world_size=1)
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    r.Init(film, scene, prt.Camera(width=W, height=H))
    for overlap in (True, False):
        film.Clear()
        r.frame_index = 0
        g = prt.dist.FilmGather(r, "cuda:0", overlap=overlap, always_collective=True)
        assert g.collective and g.backend == "nccl" and g.overlap == overlap
        for _ in range(spp):  # a gather per frame, each overlapping the next frame's rendering
            r.render_async(1)
            out = g()
        g.wait()
        torch.cuda.synchronize()
        r.synchronize()
        rgb, wts = out
        r.download()
        assert np.array_equal(rgb.cpu().numpy().reshape(H, W, 3), film.accum), overlap
        assert np.array_equal(wts.cpu().numpy().reshape(H, W), film.weights) and (film.weights == spp).all()
    ones = torch.ones(1, dtype=torch.int64, device="cuda:0")
    dist.all_reduce(ones, op=dist.ReduceOp.SUM)
    rays = torch.tensor([int(r.stats().rays_total)], dtype=torch.int64, device="cuda:0")
    dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    assert int(ones.item()) == 1 and int(rays.item()) == int(r.stats().rays_total) > 0
    dist.barrier()
    dist.destroy_process_group()
    print("NCCL_WORLD1_OK", int(rays.item()))


if __name__ == "__main__":
    main()
