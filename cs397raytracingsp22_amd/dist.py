"""One process per GPU: image tiles sharded over ranks, one gather per frame.

Partition: the image is cut into 32x32-pixel tiles, numbered row-major over a
grid whose row length is coprime with the rank count (tile_grid);
tile t belongs to rank t % world and lands in slot t // world of that rank's compact
buffer [tiles_padded][1024][3] f32.  Interleaving spreads the expensive region (teapot,
glass) over all ranks.  The RNG is keyed by the GLOBAL pixel index, so the assembled
image is bit-identical for every world size.

Per frame there is exactly ONE collective: a gather of the compact buffers to rank 0
(torch.distributed; backend "nccl" is RCCL over xGMI on ROCm — each peer reaches rank 0
over its own point-to-point link).  Rank 0 then un-permutes (K3) and tone-maps (K4).
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import abi

TILE = abi.MI_TILE
TILE_PIXELS = TILE * TILE


def tile_grid(width: int, height: int, world: int = 1):
    """(row length, rows, tiles) of the partition's tile numbering.  The row length is the image's tile columns rounded up to the
    next integer coprime with `world` (mi_rt.cpp tile_counts): a rank then walks through every column class instead of the same
    few in every row.  The surplus columns hold no pixel."""
    import math
    tx = (width + TILE - 1) // TILE
    while math.gcd(tx, world) != 1:
        tx += 1
    ty = (height + TILE - 1) // TILE
    return tx, ty, tx * ty


def tiles_padded(width: int, height: int, world: int) -> int:
    _, _, total = tile_grid(width, height, world)
    return (total + world - 1) // world


def tiles_of_rank(width: int, height: int, rank: int, world: int):
    _, _, total = tile_grid(width, height, world)
    return list(range(rank, total, world))


def compact_index(width: int, height: int, world: int):
    """For every pixel (y, x): (rank, flat index into that rank's [tiles_padded*1024] buffer).
    Pure arithmetic mirror of K3's mapping (csrc/pt_kernels.hip fb_unpermute)."""
    tx, _, _ = tile_grid(width, height, world)
    y, x = np.mgrid[0:height, 0:width]
    tile = (y // TILE) * tx + (x // TILE)
    rank = tile % world
    slot = tile // world
    idx = slot * TILE_PIXELS + (y % TILE) * TILE + (x % TILE)
    return rank, idx


def gather_compact(local, world: int, rank: int, dst: int = 0, group=None):
    """The frame's single collective.  `local` is this rank's compact buffer (torch tensor
    [tiles_padded, 1024, C]); returns [world, tiles_padded, 1024, C] on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return local.unsqueeze(0)
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal only (two ranks sharing one GPU): gloo has no device gather, stage through the host
        host = gather_compact(local.cpu(), world, rank, dst, group)
        return host.to(local.device) if host is not None else None
    if rank == dst:
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.gather(local, gather_list=[out[r] for r in range(world)], dst=dst, group=group)
        return out
    dist.gather(local, gather_list=None, dst=dst, group=group)
    return None


class TiledRenderer:
    """Per-rank frame pipeline on one GPU: K1 on this rank's tiles -> gather -> (rank 0) K3 + K4.
    torch supplies device memory, the stream and the collective; all compute is the HIP library."""

    def __init__(self, ctx, camera, rank: int = 0, world: int = 1, device: Optional[str] = None,
                 variant: int = abi.MI_VARIANT_DEFAULT, flags: int = 0, max_state_bytes: int = 0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("torch cannot see the GPU: import torch BEFORE creating a Context "
                               "(its bundled HIP runtime must be the first one loaded into the process)")
        self.torch = torch
        self.ctx, self.cam, self.rank, self.world, self.variant = ctx, camera, rank, world, variant
        self.flags, self.max_state_bytes = flags, max_state_bytes
        self.device = torch.device(device if device is not None else f"cuda:{ctx.device}")
        W, H = camera.screen_width, camera.screen_height
        self.padded = tiles_padded(W, H, world)
        self.compact = torch.zeros((self.padded, TILE_PIXELS, 3), dtype=torch.float32, device=self.device)
        self.image = self.u8 = None
        if rank == 0:
            self.image = torch.empty((H, W, 3), dtype=torch.float32, device=self.device)
            self.u8 = torch.empty((H, W, 3), dtype=torch.uint8, device=self.device)
        self.kernel_ms = []
        ctx.reserve(camera, world, max_state_bytes)          # HBM for the wavefront path state: allocated here, not in the first frame

    def render_frame(self, seed: int = 1, time_kernel: bool = False):
        torch = self.torch
        stream = torch.cuda.current_stream(self.device).cuda_stream
        st = self.ctx.render_tiles_device(self.cam, self.compact.data_ptr(), None, seed=seed, rank=self.rank,
                                          world=self.world, stream=stream, variant=self.variant, flags=self.flags,
                                          max_state_bytes=self.max_state_bytes)
        if time_kernel:
            self.kernel_ms.append(self.ctx.last_kernel_ms())
        gathered = gather_compact(self.compact, self.world, self.rank)
        if self.rank == 0:
            self.ctx.unpermute_device(self.cam, self.world, gathered.data_ptr(), self.image.data_ptr(), stream=stream)
            self.ctx.tonemap_device(self.cam, self.image.data_ptr(), self.u8.data_ptr(), stream=stream)
        return st
