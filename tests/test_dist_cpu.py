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
    for (w, h) in ((64, 48), (70, 45), (800, 600), (1280, 720)):
        for world in (1, 2, 3, 8):
            om = vd.owner_map(world, w, h)
            assert om.shape == (h, w) and om.min() == 0 and om.max() == world - 1
            cover = sum(vd.owned_mask(r, world, w, h).astype(int) for r in range(world))
            assert np.all(cover == 1)
            tx, ty = vd.tile_grid(w, h)
            assert sum(len(vd.owned_tiles(r, world, w, h)) for r in range(world)) == tx * ty


def test_tile_deal_matches_the_library_and_is_not_stripes():
    """dist.py restates vp_tile_owner (the mapping render_k uses); at 1280x720 x 8 ranks -- 160 tiles per row, where
    `tile % 8` would hand each rank vertical 8-pixel stripes -- no rank owns a whole tile column or two equal rows."""
    import volpath
    from volpath import dist as vd
    for (w, h, world) in ((1280, 720, 8), (800, 600, 8), (70, 45, 3), (64, 48, 2)):
        m = vd.tile_owner_map(world, w, h)
        ty, tx = m.shape
        lib = np.array([[volpath.tile_owner(i, j, world) for i in range(tx)] for j in range(0, ty, 7)])
        assert np.array_equal(lib, m[::7])
        # within a row: every world-th tile
        assert np.all((m[:, 1:] - m[:, :-1]) % world == 1)
    m = vd.tile_owner_map(8, 1280, 720)
    assert all(len(set(m[:, c])) == 8 for c in range(m.shape[1]))           # every column sees every rank
    shifts = vd.row_shift(np.arange(m.shape[0]), 8)
    assert np.bincount(shifts.astype(int), minlength=8).min() >= m.shape[0] // 8 - 4   # row shifts cover all residues evenly


def test_tile_deal_balances_the_julia_image():
    """weight = scatters+1 per pixel (the oracle's heat channel); contiguous bands give max/mean 2.36.
    The deal keeps it <= 1.05 at 1280x720 over 8 ranks (BASELINE config 5) and at 800x600."""
    import oracle_lib as O
    from volpath import dist as vd
    g = O.julia(64)
    sc = O.OracleScene(g, scenes.synthetic_env(), scenes.DEFAULT_SUN_DIR, scenes.DEFAULT_SUN_POWER)
    for (w, h) in ((1280, 720), (800, 600)):
        P = O.default_param(w // 4, h // 4)  # heat map at quarter resolution, one tile = 2x2 of its pixels
        acc = None
        for f in range(4):
            acc, _ = sc.render_frame(P, f, acc)
        wgt = np.repeat(np.repeat(acc[..., 3] + 4, 4, axis=0), 4, axis=1)[:h, :w]
        for world in (2, 4, 8):
            b, per = vd.balance(wgt, world)
            assert b <= 1.05, (w, h, world, b, per)
        # the degenerate deal this replaces: tile % 8 at 160 tiles per row = vertical stripes
        tx, ty = vd.tile_grid(w, h)
        t = (np.arange(h)[:, None] // 8) * tx + (np.arange(w)[None, :] // 8)
        old = np.array([wgt[(t % 8) == r].sum() for r in range(8)])
        if w == 1280:
            assert old.max() / old.mean() > b  # stripes are worse than the hashed rows


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


def test_eight_rank_reduce_is_bit_identical():
    """BASELINE configs[4]'s rank count on the CPU: eight gloo ranks, each holding the oracle's frame restricted to ITS tiles of the
    hashed row-shifted deal (vp_tile_owner's Python mirror), one reduce to rank 0: the sum of eight disjoint accumulators is the
    one-rank image bit for bit.  (The GPU side of the same count: eight contexts of one process, tests/test_c4_gpu.py.)"""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    ok = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
    assert ok


def _worker_frames(rank, world, port, q):
    import sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cuda-volpath_amd"))
    from volpath import dist as vd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # partial images of very different magnitudes: the order of the additions shows in the bits
    parts = [np.random.default_rng(5 + r).random((48, 64, 4), dtype=np.float32) * np.float32(10.0 ** (3 * r)) for r in range(world)]
    total = vd.gather_sum_in_rank_order(torch.from_numpy(parts[rank].copy()), dst=0)
    if rank == 0:
        want = parts[0].copy()
        for p in parts[1:]:
            want = want + p
        other = parts[-1].copy()
        for p in parts[-2::-1]:
            other = other + p
        q.put((bool(np.array_equal(total.numpy(), want)), bool(np.array_equal(want, other))))
    dist.barrier()
    dist.destroy_process_group()


def test_frame_split_adds_partial_images_in_rank_order():
    """bench.py --split frames over gloo, three ranks: rank 0 gets ((p0 + p1) + p2) bit for bit -- and that is not the sum in the
    opposite order, which is why the order is part of the definition."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_frames, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    same, order_free = q.get(timeout=90)
    for p in procs:
        p.join(timeout=60)
    assert same and not order_free
