"""CPU suite, part 3: the N>1 path (pixel-tile sharding + one reduce) with world_size 2 over gloo.
The per-rank renderer here is the CPU oracle standing in for the GPU; what is under test is the
tile->rank mapping and the reduce giving a bit-identical image."""
import os
import socket

import numpy as np
import pytest

import scenes


def test_tile_partition_is_disjoint_and_complete():
    from volpath import dist as vd
    for (w, h) in ((64, 48), (70, 45), (800, 600)):
        for world in (1, 2, 3, 8):
            om = vd.owner_map(world, w, h)
            assert om.shape == (h, w) and om.min() == 0 and om.max() == min(world - 1, om.max())
            cover = sum(vd.owned_mask(r, world, w, h).astype(int) for r in range(world))
            assert np.all(cover == 1)
            tx, ty = vd.tile_grid(w, h)
            assert sum(len(vd.owned_tiles(r, world, w, h)) for r in range(world)) == tx * ty


def test_round_robin_tiles_balance_the_julia_image():
    """weight = scatters+1 per pixel; round-robin 8x8 tiles keep max/mean near 1 (bands: 2.36)."""
    import oracle_lib as O
    from volpath import dist as vd
    g = O.julia(64)
    sc = O.OracleScene(g, scenes.synthetic_env(), scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER)
    P = O.default_param(200, 150)
    acc, _ = sc.render_frame(P, 0)
    wgt = acc[..., 3] + 1
    loads = np.array([wgt[vd.owned_mask(r, 8, 200, 150)].sum() for r in range(8)])
    assert loads.max() / loads.mean() < 1.25


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    import oracle_lib as O
    from volpath import dist as vd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H = 40, 24
    g = O.julia(32)
    sc = O.OracleScene(g, scenes.synthetic_env(), scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER,
                       rng_mode=O.RNG_PHILOX, seed=(4, 2))
    P = O.default_param(W, H)
    full = None
    for f in range(3):
        full, _ = sc.render_frame(P, f, full)
    mine = np.where(vd.owned_mask(rank, world, W, H)[..., None], full, 0).astype(np.float32)
    acc = torch.from_numpy(mine.copy())
    vd.reduce_accumulator(acc, dst=0)
    if rank == 0:
        q.put(bool(np.array_equal(acc.numpy(), full)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_reduce_is_bit_identical():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
    assert ok
