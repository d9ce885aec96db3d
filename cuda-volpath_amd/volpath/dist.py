"""Pixel-tile sharding across the GPUs of one node.

Every (pixel, sample) is independent and all scene data is read-only, so the path shards with no
data-path collective: the 8x8 pixel tiles are dealt round-robin (tile t -> rank t % world, the same
mapping render_k uses), each rank accumulates only its tiles into a full-frame buffer that stays
zero elsewhere, and ONE reduce(sum) of the HDR accumulators to rank 0 ends the render (RCCL over
xGMI on GPUs; gloo in the CPU tests).  Tiles are disjoint, so the sum adds one non-zero term and
zeros per pixel: the N-GPU image is bit-identical to the 1-GPU image.
"""
import numpy as np

TILE = 8


def tile_grid(width, height):
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


def owned_tiles(rank, world, width, height):
    tx, ty = tile_grid(width, height)
    return np.arange(rank, tx * ty, world)


def owner_map(world, width, height):
    """rank that owns each pixel, shape (H, W)."""
    tx, ty = tile_grid(width, height)
    t = (np.arange(height)[:, None] // TILE) * tx + (np.arange(width)[None, :] // TILE)
    return (t % world).astype(np.int32)


def owned_mask(rank, world, width, height):
    return owner_map(world, width, height) == rank


def reduce_accumulator(acc, dst=0, group=None):
    """Sum the per-rank HDR accumulators onto rank `dst` (torch tensor, in place)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(acc, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return acc
