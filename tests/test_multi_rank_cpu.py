"""The N>1 path on CPU: 2 processes over gloo.  Each rank renders ITS strip (with the CPU oracle standing in for the
GPU context, which needs a GPU), then the product's shard helpers (tinyrenderder_amd/shard.py — the same code
bench.py runs over RCCL) join the strips and reduce the counters.  Every rank must end with the whole frame."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
from tinyrenderder_amd import shard


def _worker(rank, world, port, name, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        case = cases.CASES[name]()
        W, H, bpp = case["width"], case["height"], case["bpp"]
        y0, y1 = shard.strip_rows(H, world, rank)
        fb, z, st = cases.run_oracle(case, strip=(y0, y1))
        fb[:y0] = 0xAA; fb[y1:] = 0xAA                       # rows a rank does not own hold garbage before the gather
        full = torch.from_numpy(fb.reshape(-1))
        shard.gather_strips(full, W, H, bpp, rank, world)
        zfull = torch.from_numpy(z.view(np.uint8).reshape(-1))
        shard.gather_strips(zfull, W, H, 8, rank, world)
        merged = shard.reduce_stats(st)
        ret[rank] = (full.numpy().reshape(H, W, bpp).copy(), zfull.numpy().view(np.float64).reshape(H, W).copy(), merged)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["flat_persp_512", "phong_512"])
def test_two_ranks_gloo_strips_equal_whole_frame(name):
    world, port = 2, 29500 + (os.getpid() % 2000)
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, name, ret), nprocs=world, join=True)
    fb, z, st = cases.run_oracle(cases.CASES[name]())
    for rank in range(world):
        rfb, rz, rst = ret[rank]
        assert np.array_equal(rfb, fb), f"rank {rank}: gathered framebuffer differs from the single-rank frame"
        assert np.array_equal(rz.view(np.uint64), z.view(np.uint64))
        assert rst == st


def _band_worker(rank, world, port, band, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        case = cases.CASES["flat_persp_512"]()
        W, H, bpp = case["width"], case["height"], case["bpp"]
        fb, z, _ = cases.run_oracle(case)
        own = np.zeros(H, bool)
        for y0, y1 in shard.band_rows_of(H, world, rank, band):
            own[y0:y1] = True
        fb[~own] = 0xAA                                          # rows of other ranks hold garbage before the gather
        zb = z.view(np.uint8).reshape(H, W * 8).copy(); zb[~own] = 0xAA
        full = torch.from_numpy(fb.reshape(-1))
        shard.gather_bands(full, W, H, bpp, band, rank, world)
        zfull = torch.from_numpy(zb.reshape(-1))
        work = shard.gather_bands(zfull, W, H, 8, band, rank, world, async_op=True)       # the asynchronous form bench.py uses
        work.wait()
        ret[rank] = (full.numpy().reshape(H, W, bpp).copy(), zfull.numpy().view(np.float64).reshape(H, W).copy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,band", [(2, 64), (4, 32), (4, 128)])
def test_interleaved_bands_gather_gloo(world, band):
    """shard.gather_bands: one in-place all-gather per period of world * band rows joins interleaved bands (world 2 and 4)."""
    port = 31500 + (os.getpid() % 2000) + world
    ret = mp.Manager().dict()
    mp.spawn(_band_worker, args=(world, port, band, ret), nprocs=world, join=True)
    fb, z, _ = cases.run_oracle(cases.CASES["flat_persp_512"]())
    for rank in range(world):
        rfb, rz = ret[rank]
        assert np.array_equal(rfb, fb), f"rank {rank}: gathered framebuffer differs"
        assert np.array_equal(rz.view(np.uint64), z.view(np.uint64))
    with pytest.raises(ValueError):
        shard.band_rows_of(500, 4, 0, 32)


def test_strip_rows_partition():
    for H, G in ((4096, 8), (8192, 8), (512, 2), (96, 3)):
        edges = [shard.strip_rows(H, G, r) for r in range(G)]
        assert edges[0][0] == 0 and edges[-1][1] == H and all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
    with pytest.raises(ValueError):
        shard.strip_rows(100, 8, 0)
