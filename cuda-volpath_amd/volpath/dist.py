"""Pixel-tile sharding across the GPUs of one node.

Every (pixel, sample) is independent and all scene data is read-only, so the path shards with no
data-path collective: the 8x8 pixel tiles are dealt to the ranks (tile (tx, ty) -> rank
(tx + row_shift(ty)) % world: every world-th tile of a tile row, the rows shifted against each other
by a hash of the row index -- the mapping render_k uses, vp_kernels.h owned_tile), each rank
accumulates only its tiles into a full-frame buffer that stays zero elsewhere, and ONE reduce(sum)
of the HDR accumulators to rank 0 ends the render (RCCL over xGMI on GPUs; gloo in the CPU tests).
Tiles are disjoint, so the sum adds one non-zero term and zeros per pixel: the N-GPU image is
bit-identical to the 1-GPU image.
"""
import numpy as np

TILE = 8


def tile_grid(width, height):
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


def row_shift(ty, world):
    """vp_kernels.h tile_row_shift: ((ty * 0x9E3779B1) mod 2^32 >> 15) % world"""
    ty = np.asarray(ty, np.uint64)
    return (((ty * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)) >> np.uint64(15)) % np.uint64(world)


def tile_owner_map(world, width, height):
    """rank that owns each tile, shape (tiles_y, tiles_x)."""
    tx, ty = tile_grid(width, height)
    return ((np.arange(tx, dtype=np.uint64)[None, :] + row_shift(np.arange(ty), world)[:, None]) % np.uint64(world)).astype(np.int32)


def owned_tiles(rank, world, width, height):
    """row-major tile indices of one rank"""
    m = tile_owner_map(world, width, height)
    return np.flatnonzero(m.ravel() == rank)


def owner_map(world, width, height):
    """rank that owns each pixel, shape (H, W)."""
    m = tile_owner_map(world, width, height)
    return np.repeat(np.repeat(m, TILE, axis=0), TILE, axis=1)[:height, :width]


def owned_mask(rank, world, width, height):
    return owner_map(world, width, height) == rank


def balance(weight, world):
    """max/mean over ranks of the summed per-pixel weight (H, W) under this deal"""
    h, w = weight.shape
    own = owner_map(world, w, h)
    per = np.array([weight[own == r].sum() for r in range(world)], np.float64)
    return float(per.max() / per.mean()), per


def reduce_accumulator(acc, dst=0, group=None):
    """Sum the per-rank HDR accumulators onto rank `dst` (torch tensor, in place)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(acc, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return acc


def gather_sum_in_rank_order(part, dst=0, group=None):
    """The frame split of a fixed job (bench.py --split frames): every rank holds a partial image of ALL pixels
    (its share of the frames); rank `dst` gathers them and returns ((p0 + p1) + p2) + ... -- a defined result, because binary32
    addition does not associate (a reduce would add in the library's order).  Other ranks return None."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    if world == 1:
        return part
    rank = dist.get_rank(group)
    parts = [torch.empty_like(part) for _ in range(world)] if rank == dst else None
    dist.gather(part, parts, dst=dst, group=group)
    if rank != dst:
        return None
    total = parts[0].clone()
    for q in parts[1:]:
        total = total + q
    return total
