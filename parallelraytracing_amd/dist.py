"""Multi-GPU plumbing: one process per GPU, image tiled across ranks, one gather of per-tile radiance.

The reference is single-GPU (cudaSetDevice(0) only, src/backend/optix/renderer.cpp:217); this is the new part
the north star asks for.  Pixels are independent, so the path shards with NO data-path collective: every
rank renders the 8x8 tiles t with t % world == rank (scene and BVH replicated), and ONE gather per frame
brings the tile-ordered payloads [stride][r,g,b,weight] to rank 0, which un-tiles them into the Film layout
(prt_film_resolve).  torch.distributed ("nccl" == RCCL over xGMI on ROCm, "gloo" on CPU) is only the
transport.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


# ---- tile layout (must match tile_pixel() / k_resolve in csrc/prt_kernels.hip) --------------------------------
def tile_layout(W: int, H: int, world: int) -> Tuple[int, int, int]:
    """tiles_x, tiles_y, stride (pixels of one rank's payload, equal on all ranks)."""
    tx, ty = (W + 7) // 8, (H + 7) // 8
    stride = ((tx * ty + world - 1) // world) * 64
    return tx, ty, stride


def pixel_owner_and_slot(W: int, H: int, world: int):
    """For every pixel (row-major): owning rank and slot inside that rank's payload."""
    tx, _, _ = tile_layout(W, H, world)
    y, x = np.mgrid[0:H, 0:W]
    gt = (y // 8) * tx + (x // 8)
    rank = gt % world
    slot = (gt // world) * 64 + (y % 8) * 8 + (x % 8)
    return rank.astype(np.int64), slot.astype(np.int64)


def pack_tiles_numpy(accum: np.ndarray, weights: np.ndarray, rank: int, world: int) -> np.ndarray:
    """Host restatement of one rank's payload (what prt_film_local points at), for CPU tests."""
    H, W = weights.shape
    _, _, stride = tile_layout(W, H, world)
    owner, slot = pixel_owner_and_slot(W, H, world)
    out = np.zeros((stride, 4), np.float32)
    m = owner == rank
    out[slot[m], :3] = accum[m]
    out[slot[m], 3] = weights[m]
    return out


def untile_numpy(gathered: np.ndarray, W: int, H: int) -> Tuple[np.ndarray, np.ndarray]:
    """Host restatement of prt_film_resolve: gathered is [world, stride, 4]."""
    world = gathered.shape[0]
    owner, slot = pixel_owner_and_slot(W, H, world)
    px = gathered[owner, slot]
    return np.ascontiguousarray(px[..., :3]), np.ascontiguousarray(px[..., 3])


# ---- device side ----------------------------------------------------------------------------------------------------
class DevicePointer:
    """Zero-copy view of a raw device pointer for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr: int, n_floats: int):
        self.__cuda_array_interface__ = {"shape": (int(n_floats),), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def local_payload_tensor(renderer, device):
    import torch
    ptr, n = renderer.film_local()
    return torch.as_tensor(DevicePointer(ptr, n), device=device)


class FilmGather:
    """Per-frame gather of the ranks' payloads to rank 0 and un-tiling into (rgb_sum, weight) tensors.

    overlap = False (default): gather and un-tiling are enqueued on the CURRENT stream; the returned tensors can be consumed
    on that stream right away (tonemap, read-back), as any torch result.
    overlap = True (opt-in; what bench.py uses): the frame's payload is SNAPSHOT on the render stream (one device-to-device copy of the
    rank's tiles, 4 MB per GPU at 1080p / N = 8) and the gather + un-tiling run on a side stream from that snapshot, so
    the render stream goes straight on with the next frame (the gather is ~0.15-0.3 ms of an 8 ms step at N = 8).
    With overlap the result tensors are written on the SIDE stream: a consumer on the current stream must call `wait()`
    first (or torch.cuda.synchronize()); reading them without it races with the gather.
    always_collective = True runs the gather even with one rank (a 1-rank RCCL gather: the -m gpu test that makes
    init_process_group("nccl"), the zero-copy payload tensor and dist.gather execute on real hardware)."""

    def __init__(self, renderer, device, group=None, overlap=False, always_collective=False):
        import torch
        import torch.distributed as dist
        self.r = renderer
        self.device = device
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.collective = dist.is_initialized() and (self.world > 1 or always_collective)
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        self.local = local_payload_tensor(renderer, device)
        n = self.local.numel()
        W, H = renderer.film.width, renderer.film.height
        self.overlap = bool(overlap) and self.collective and self.backend != "gloo"
        self.side = torch.cuda.Stream(device=device) if self.overlap else None
        self.snap = torch.empty_like(self.local) if self.overlap else None
        if self.rank == 0:
            self.gathered = torch.empty(self.world * n, dtype=torch.float32, device=device)
            self.parts = list(self.gathered.view(self.world, n).unbind(0)) if self.collective else None
            self.rgb = torch.empty(H * W * 3, dtype=torch.float32, device=device)
            self.weight = torch.empty(H * W, dtype=torch.float32, device=device)
        else:
            self.gathered = self.parts = self.rgb = self.weight = None

    def wait(self):
        """Make the current stream wait for the gather / un-tiling enqueued by the last call."""
        import torch
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)

    def __call__(self):
        """Enqueue gather + resolve; returns (rgb, weight) on rank 0, else None (see `wait`)."""
        import torch
        import torch.distributed as dist
        if not self.collective:
            if self.rank != 0:
                return None
            self.r.film_resolve(self.local.data_ptr(), self.rgb.data_ptr(), self.weight.data_ptr())
            return self.rgb, self.weight
        if self.backend == "gloo":
            # rehearsal path (several ranks sharing one GPU, or no RCCL): stage through host memory
            torch.cuda.current_stream().synchronize()
            host = self.local.cpu()
            parts = [torch.empty_like(host) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(host, parts, dst=0, group=self.group)
            if self.rank != 0:
                return None
            self.gathered.copy_(torch.cat(parts))
            self.r.film_resolve(self.gathered.data_ptr(), self.rgb.data_ptr(), self.weight.data_ptr())
            return self.rgb, self.weight
        if not self.overlap:
            dist.gather(self.local, self.parts if self.rank == 0 else None, dst=0, group=self.group)
            if self.rank != 0:
                return None
            self.r.film_resolve(self.gathered.data_ptr(), self.rgb.data_ptr(), self.weight.data_ptr())
            return self.rgb, self.weight
        cur = torch.cuda.current_stream()
        cur.wait_stream(self.side)      # the previous frame's gather has finished reading the snapshot
        self.snap.copy_(self.local)     # on the render stream, behind this frame's last k_accumulate
        self.side.wait_stream(cur)
        with torch.cuda.stream(self.side):
            dist.gather(self.snap, self.parts if self.rank == 0 else None, dst=0, group=self.group)
            if self.rank == 0:
                self.r.film_resolve(self.gathered.data_ptr(), self.rgb.data_ptr(), self.weight.data_ptr(),
                                    stream=self.side.cuda_stream)
        return (self.rgb, self.weight) if self.rank == 0 else None
