"""Multi-process CPU tests (gloo, world_size 2 and 8) of the frame / tile sharding logic with a stand-in
for the network (nearest x4 upscale, for which halo tiling is exact), plus geometry unit tests."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from superresolution_amd import sharding, synth


def fake_forward(img):  # u8 HWC (or NHWC) -> u8 x4 (nearest)
    d = img.dim() - 3
    return img.repeat_interleave(4, dim=d).repeat_interleave(4, dim=d + 1)


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 9, 64):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                seen += list(sharding.shard_range(n, r, world))
            assert seen == list(range(n))
            sizes = [len(sharding.shard_range(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_range(4, 2, 2)


def test_tile_grid_and_stitch():
    h, w = 67, 101
    tiles = sharding.tile_grid(h, w, 32, 48, 10)
    cover = np.zeros((h, w), int)
    for t in tiles:
        cover[t.y0:t.y1, t.x0:t.x1] += 1
        assert t.py0 <= t.y0 and t.py1 >= t.y1 and t.px0 <= t.x0 and t.px1 >= t.x1
        assert t.py0 == max(t.y0 - 10, 0) and t.px1 == min(t.x1 + 10, w)
    assert (cover == 1).all()
    img = torch.from_numpy(synth.random_image(h, w, 1))
    out = sharding.upscale_tiled(fake_forward, img, 4, 32, 48, 10)
    assert torch.equal(out, fake_forward(img))
    # tiles of equal crop shape stacked into one call: same frame, fewer calls
    calls = []

    def counting(x):
        calls.append(tuple(x.shape))
        return fake_forward(x)

    assert torch.equal(sharding.upscale_tiled_batched(counting, img, 4, 32, 48, 10), fake_forward(img))
    assert all(len(c) == 4 for c in calls) and sum(c[0] for c in calls) == len(tiles)
    calls.clear()
    big = torch.from_numpy(synth.random_image(128, 192, 2))          # 4 x 4 grid: 4 crop shapes (corner, two kinds of edge, interior) for 16 tiles
    assert torch.equal(sharding.upscale_tiled_batched(counting, big, 4, 32, 48, 10), fake_forward(big))
    assert len(calls) == 4 and all(c[0] == 4 for c in calls)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        img = torch.from_numpy(synth.random_image(70, 90, 3))
        out = sharding.upscale_tiled_sharded(fake_forward, img, 4, 32, 32, 6, rank, world)
        out_b = sharding.upscale_tiled_sharded(fake_forward, img, 4, 32, 32, 6, rank, world, batched=True)
        frames = torch.from_numpy(synth.random_image(16, 20, 4, n=5))
        own, outs = sharding.upscale_frames_sharded(fake_forward, frames, rank, world)
        ok = all(torch.equal(o, fake_forward(frames[i])) for i, o in zip(own, outs))
        if rank == 0:
            ok = ok and torch.equal(out, fake_forward(img)) and torch.equal(out_b, out)
        else:
            ok = ok and out is None and out_b is None
        ret[rank] = (bool(ok), list(own))
    finally:
        dist.destroy_process_group()


def test_tile_and_frame_sharding_world2_gloo():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret[0][0] and ret[1][0]
    assert ret[0][1] + ret[1][1] == [0, 1, 2, 3, 4]


def test_tile_assignment_of_the_1080p_grid_on_8_ranks():
    """Tile mode of one 1920x1080 frame as the benchmark runs it (2 x 4 grid of 480x540 tiles + 32-pixel halo): on
    eight ranks every rank owns exactly one tile, on 1 / 2 / 4 ranks equal shares, each tile exactly once."""
    tiles = sharding.tile_grid(1080, 1920, 540, 480, 32)
    assert len(tiles) == 8
    for world in (1, 2, 4, 8):
        owned = [list(sharding.shard_range(len(tiles), r, world)) for r in range(world)]
        assert sorted(sum(owned, [])) == list(range(8))
        assert all(len(o) == 8 // world for o in owned)
    assert [list(sharding.shard_range(8, r, 8)) for r in range(8)] == [[i] for i in range(8)]


def _worker8(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # a small frame cut 2 x 4 like the 1080p grid: 8 tiles on 8 ranks, one each, gathered to rank 0
        img = torch.from_numpy(synth.random_image(64, 128, 5))
        assert len(sharding.tile_grid(64, 128, 32, 32, 4)) == 8
        out = sharding.upscale_tiled_sharded(fake_forward, img, 4, 32, 32, 4, rank, world, batched=True)
        frames = torch.from_numpy(synth.random_image(8, 12, 6, n=19))     # 19 frames: shares of 3 and 2
        own, outs = sharding.upscale_frames_sharded(fake_forward, frames, rank, world)
        ok = all(torch.equal(o, fake_forward(frames[i])) for i, o in zip(own, outs))
        ok = ok and (torch.equal(out, fake_forward(img)) if rank == 0 else out is None)
        ret[rank] = (bool(ok), list(own))
    finally:
        dist.destroy_process_group()


def test_tile_and_frame_sharding_world8_gloo():
    world = 8
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker8, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret[r][0] for r in range(world))
    assert sum((ret[r][1] for r in range(world)), []) == list(range(19))
    assert sorted(len(ret[r][1]) for r in range(world)) == [2] * 5 + [3] * 3
