"""Multi-GPU work partitioning: frames (or halo-overlapped tiles of one large frame) are
independent units, so every rank runs the same single-GPU engine on its share and there is no
collective on the data path.  The only exchange is the optional final gather of u8 output tiles
to one rank (torch.distributed; RCCL over xGMI on GPUs, gloo in the CPU tests).

The reference has no tiling (it resizes or rejects other sizes, realesrgan.cpp:278-289); tiling
follows the usual Real-ESRGAN practice of cropping each tile with a halo, upscaling the crop on
its own and keeping the centre.  It is exact when the halo covers the network's receptive field
(about 15*num_block + 4 low-resolution pixels for RRDBNet) and approximate otherwise; RCAN cannot
be tiled exactly at all because its channel attention pools over the whole frame.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Sequence

import torch


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced share of range(n_items) for `rank` (sizes differ by at most one)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


@dataclass(frozen=True)
class Tile:
    y0: int
    y1: int
    x0: int
    x1: int      # core region of the low-resolution frame this tile is responsible for
    py0: int
    py1: int
    px0: int
    px1: int     # crop actually fed to the network (core + halo, clipped to the frame)


def tile_grid(h: int, w: int, tile_h: int, tile_w: int, halo: int) -> List[Tile]:
    """Row-major grid of tiles covering an h x w frame."""
    if min(h, w, tile_h, tile_w) <= 0 or halo < 0:
        raise ValueError("bad tile geometry")
    tiles = []
    for y0 in range(0, h, tile_h):
        for x0 in range(0, w, tile_w):
            y1, x1 = min(y0 + tile_h, h), min(x0 + tile_w, w)
            tiles.append(Tile(y0, y1, x0, x1, max(y0 - halo, 0), min(y1 + halo, h), max(x0 - halo, 0),
                              min(x1 + halo, w)))
    return tiles


def upscale_tile(forward_u8: Callable[[torch.Tensor], torch.Tensor], img: torch.Tensor, t: Tile, scale: int):
    """Upscale one tile's crop and cut its core out of the result.  img: u8 HWC."""
    out = forward_u8(img[t.py0:t.py1, t.px0:t.px1].contiguous())
    oy, ox = (t.y0 - t.py0) * scale, (t.x0 - t.px0) * scale
    return out[oy:oy + (t.y1 - t.y0) * scale, ox:ox + (t.x1 - t.x0) * scale]


def stitch(tiles: Sequence[Tile], pieces: Sequence[torch.Tensor], h: int, w: int, scale: int) -> torch.Tensor:
    out = torch.empty((h * scale, w * scale, 3), dtype=torch.uint8, device=pieces[0].device)
    for t, p in zip(tiles, pieces):
        out[t.y0 * scale:t.y1 * scale, t.x0 * scale:t.x1 * scale] = p
    return out


def upscale_tiled(forward_u8, img: torch.Tensor, scale: int, tile_h: int, tile_w: int, halo: int) -> torch.Tensor:
    """Single-process tiled upscale of one frame."""
    h, w = img.shape[:2]
    tiles = tile_grid(h, w, tile_h, tile_w, halo)
    return stitch(tiles, [upscale_tile(forward_u8, img, t, scale) for t in tiles], h, w, scale)


def upscale_tiles_batched(forward_u8, img: torch.Tensor, tiles: Sequence[Tile], scale: int) -> List[torch.Tensor]:
    """Upscaled cores of `tiles`, in order.  Tiles whose crops have the same shape (interior tiles; edge tiles
    of one kind) are stacked and go through the network in one call, so that the engine's stream lanes overlap
    them; every frame of a batch is computed exactly as it would be alone, so the result equals
    [upscale_tile(...) for t in tiles]."""
    groups = {}
    for i, t in enumerate(tiles):
        groups.setdefault((t.py1 - t.py0, t.px1 - t.px0), []).append(i)
    pieces = [None] * len(tiles)
    for idx in groups.values():
        crops = torch.stack([img[tiles[i].py0:tiles[i].py1, tiles[i].px0:tiles[i].px1] for i in idx])
        outs = forward_u8(crops)
        for j, i in enumerate(idx):
            t = tiles[i]
            oy, ox = (t.y0 - t.py0) * scale, (t.x0 - t.px0) * scale
            pieces[i] = outs[j][oy:oy + (t.y1 - t.y0) * scale, ox:ox + (t.x1 - t.x0) * scale]
    return pieces


def upscale_tiled_batched(forward_u8, img: torch.Tensor, scale: int, tile_h: int, tile_w: int, halo: int) -> torch.Tensor:
    """upscale_tiled with the tiles of equal crop shape batched (forward_u8 must accept uint8 [N,H,W,3])."""
    h, w = img.shape[:2]
    tiles = tile_grid(h, w, tile_h, tile_w, halo)
    return stitch(tiles, upscale_tiles_batched(forward_u8, img, tiles, scale), h, w, scale)


def upscale_tiled_sharded(forward_u8, img: torch.Tensor, scale: int, tile_h: int, tile_w: int, halo: int,
                          rank: int, world: int, gather_to: int = 0, group=None, batched: bool = False,
                          always_gather: bool = False):
    """Each rank upscales tiles rank, rank+world, ...; the u8 pieces are gathered to `gather_to`,
    which returns the stitched frame (other ranks return None).  One collective, at the end.
    batched=True sends a rank's tiles of equal crop shape through forward_u8 together ([N,H,W,3]).
    always_gather=True runs the collective for a single rank too (needs an initialised process group): the
    way to execute the RCCL gather of device buffers on a one-GPU box."""
    import torch.distributed as dist
    h, w = img.shape[:2]
    tiles = tile_grid(h, w, tile_h, tile_w, halo)
    mine = [i for i in range(len(tiles)) if i % world == rank]
    if batched:
        pieces = dict(zip(mine, upscale_tiles_batched(forward_u8, img, [tiles[i] for i in mine], scale)))
    else:
        pieces = {i: upscale_tile(forward_u8, img, tiles[i], scale) for i in mine}
    if world == 1 and not always_gather:
        return stitch(tiles, [pieces[i] for i in range(len(tiles))], h, w, scale)
    # equal-size flat buffers so that one gather moves everything
    per_rank = (len(tiles) + world - 1) // world
    tile_bytes = tile_h * tile_w * 3 * scale * scale
    dev = img.device
    send = torch.zeros(per_rank * tile_bytes, dtype=torch.uint8, device=dev)
    for slot, i in enumerate(mine):
        flat = pieces[i].reshape(-1)
        send[slot * tile_bytes:slot * tile_bytes + flat.numel()] = flat
    # RCCL ("nccl") gathers device buffers over xGMI; gloo (CPU tests, rehearsals with fewer GPUs than ranks)
    # only moves host memory, so device pieces are staged through the host for it
    staged = dev.type == "cuda" and dist.get_backend(group) == "gloo"
    if staged:
        send = send.cpu()
    recv = [torch.empty_like(send) for _ in range(world)] if rank == gather_to else None
    dist.gather(send, recv, dst=gather_to, group=group)
    if rank != gather_to:
        return None
    if staged:
        recv = [r.to(dev) for r in recv]
    out_pieces = []
    for i, t in enumerate(tiles):
        r, slot = i % world, i // world
        n = (t.y1 - t.y0) * (t.x1 - t.x0) * 3 * scale * scale
        out_pieces.append(recv[r][slot * tile_bytes:slot * tile_bytes + n].reshape((t.y1 - t.y0) * scale,
                                                                                  (t.x1 - t.x0) * scale, 3))
    return stitch(tiles, out_pieces, h, w, scale)


def upscale_frames_sharded(forward_u8, frames: torch.Tensor, rank: int, world: int) -> tuple:
    """Frame sharding: returns (range of frame indices this rank owns, their upscaled frames).
    No communication at all."""
    own = shard_range(frames.shape[0], rank, world)
    outs = [forward_u8(frames[i]) for i in own]
    return own, outs
