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
    """Per-frame gather of the ranks' payloads to rank 0 and un-tiling into (rgb_sum, weight) tensors."""

    def __init__(self, renderer, device, group=None):
        import torch
        import torch.distributed as dist
        self.r = renderer
        self.device = device
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.local = local_payload_tensor(renderer, device)
        n = self.local.numel()
        W, H = renderer.film.width, renderer.film.height
        if self.rank == 0:
            self.gathered = torch.empty(self.world * n, dtype=torch.float32, device=device)
            self.parts = list(self.gathered.view(self.world, n).unbind(0)) if self.world > 1 else None
            self.rgb = torch.empty(H * W * 3, dtype=torch.float32, device=device)
            self.weight = torch.empty(H * W, dtype=torch.float32, device=device)
        else:
            self.gathered = self.parts = self.rgb = self.weight = None

    def __call__(self):
        """Enqueue gather + resolve on the current stream; returns (rgb, weight) on rank 0, else None."""
        import torch
        import torch.distributed as dist
        if self.world > 1:
            if dist.get_backend(self.group) == "gloo":
                # rehearsal path (several ranks sharing one GPU, or no RCCL): stage through host memory
                torch.cuda.current_stream().synchronize()
                host = self.local.cpu()
                parts = [torch.empty_like(host) for _ in range(self.world)] if self.rank == 0 else None
                dist.gather(host, parts, dst=0, group=self.group)
                if self.rank == 0:
                    self.gathered.copy_(torch.cat(parts))
            else:
                dist.gather(self.local, self.parts if self.rank == 0 else None, dst=0, group=self.group)
            src = self.gathered
        else:
            src = self.local
        if self.rank != 0:
            return None
        self.r.film_resolve(src.data_ptr(), self.rgb.data_ptr(), self.weight.data_ptr())
        return self.rgb, self.weight
