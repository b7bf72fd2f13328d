"""Directory / frame-stream driver: many images in -> upscaled images out.

The reference's only executable handles one image per process (realesrgan/mtk/cpp/jni/src/main.cpp:16-114:
load -> inference -> save, wall-clock report).  This driver keeps that flow and report but feeds a stream: files
are decoded by a small thread pool, grouped into batches of equal size, run through
Engine.forward_u8_host (pinned staging, PCIe copies and host copies overlapped with compute across the engine's
lanes) and encoded by the pool again, so the GPU never waits for a codec.  With --gpus N the file list is cut
into N contiguous shares (sharding.shard_range), one process per GPU, no communication.

  python -m superresolution_amd.batch <model> <input_dir> <output_dir> [--batch 12] [--dtype f16] [--gpus N]
                                      [--tile H W --halo P] [--suffix _x4]

<model>: a runner weight file (.srmi, export.save_weights) or an official checkpoint (.pt / .pth; family and scale
from the file name as in the reference's loaders).  Frames larger than --tile are cut into halo-overlapped tiles
(sharding.tile_grid) that go through the same batches.
"""
from __future__ import annotations

import argparse
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from typing import Iterable, Iterator, List, Tuple

import numpy as np

from . import _lib, sharding

IMAGE_EXT = (".png", ".jpg", ".jpeg", ".bmp", ".webp", ".tif", ".tiff")


def load_model(path, **engine_kwargs):
    """Weight file or checkpoint -> module of this package."""
    from . import export
    if path.lower().endswith(".srmi"):
        model = export.load_weights(path, **engine_kwargs)
    else:
        from .edsr_model import load_edsr_from_checkpoint
        from .rcan_model import infer_scale_from_model_name, load_rcan_from_checkpoint
        from .realesrgan_model import load_realesrgan_from_checkpoint
        name = os.path.basename(path).lower()
        if "edsr" in name:
            model = load_edsr_from_checkpoint(path, 4, **engine_kwargs)
        elif "rcan" in name:
            model = load_rcan_from_checkpoint(path, infer_scale_from_model_name(path), **engine_kwargs)
        else:
            model, _ = load_realesrgan_from_checkpoint(path, **engine_kwargs)
    cfg = model._engine_config()
    if cfg["arch"] != _lib.ARCH_RRDBNET and not cfg.get("with_meanshift", False):
        from .edsr_model import get_meanshift_params
        p = get_meanshift_params()
        model.engine().set_rgb_shift([m * p["rgb_range"] for m in p["rgb_mean"]])
    return model


def upscale_stream(engine, frames: Iterable[Tuple[object, np.ndarray]], batch: int = 12,
                   round_mode: int = _lib.ROUND_HALF_UP, tile: Tuple[int, int] = None, halo: int = 16
                   ) -> Iterator[Tuple[object, np.ndarray]]:
    """frames: iterable of (key, uint8 HWC array).  Yields (key, upscaled uint8 HWC array) in input order.
    Consecutive frames of equal size share a batch; a frame larger than `tile` (h, w) is cut into halo-overlapped
    tiles which are batched the same way and stitched on the host."""
    s = engine.scale
    pending: List[Tuple[object, np.ndarray]] = []

    def flush():
        if not pending:
            return
        out = engine.forward_u8_host(np.stack([f for _, f in pending]), round_mode)
        for (key, _), o in zip(pending, out):
            yield key, o
        pending.clear()

    for key, img in frames:
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape[:2]
        if tile and (h > tile[0] or w > tile[1]):
            yield from flush()
            tiles = sharding.tile_grid(h, w, tile[0], tile[1], halo)
            groups = {}
            for i, t in enumerate(tiles):      # crops of equal size share batches
                groups.setdefault((t.py1 - t.py0, t.px1 - t.px0), []).append(i)
            full = np.empty((h * s, w * s, 3), np.uint8)
            for idx in groups.values():
                for k in range(0, len(idx), batch):
                    part = idx[k:k + batch]
                    crops = np.stack([img[tiles[i].py0:tiles[i].py1, tiles[i].px0:tiles[i].px1] for i in part])
                    outs = engine.forward_u8_host(crops, round_mode)
                    for i, o in zip(part, outs):
                        t = tiles[i]
                        oy, ox = (t.y0 - t.py0) * s, (t.x0 - t.px0) * s
                        full[t.y0 * s:t.y1 * s, t.x0 * s:t.x1 * s] = o[oy:oy + (t.y1 - t.y0) * s, ox:ox + (t.x1 - t.x0) * s]
            yield key, full
            continue
        if pending and (pending[0][1].shape != img.shape or len(pending) == batch):
            yield from flush()
        pending.append((key, img))
        if len(pending) == batch:
            yield from flush()
    yield from flush()


def list_images(input_dir):
    return sorted(f for f in os.listdir(input_dir) if f.lower().endswith(IMAGE_EXT))


def run_directory(model, input_dir, output_dir, names=None, batch=12, tile=None, halo=16, suffix="", workers=4,
                  log=print):
    """Upscale the image files `names` (default: all) of input_dir into output_dir.  Returns
    (files written, output megapixels, seconds)."""
    from PIL import Image
    os.makedirs(output_dir, exist_ok=True)
    names = list_images(input_dir) if names is None else list(names)
    eng = model.engine()

    def decode(name):
        with Image.open(os.path.join(input_dir, name)) as im:
            return name, np.asarray(im.convert("RGB"), dtype=np.uint8)

    def encode(name, arr):
        stem, ext = os.path.splitext(name)
        path = os.path.join(output_dir, f"{stem}{suffix}{ext if ext.lower() in ('.png', '.bmp', '.tif', '.tiff') else '.png'}")
        Image.fromarray(arr, "RGB").save(path)
        return path

    t0 = time.perf_counter()
    mpix = 0.0
    written = []
    with ThreadPoolExecutor(workers) as pool:
        ahead = 2 * batch

        def decoded():
            futs = []
            it = iter(names)
            for n in it:
                futs.append(pool.submit(decode, n))
                if len(futs) >= ahead:
                    yield futs.pop(0).result()
            for f in futs:
                yield f.result()

        saves = []
        for name, out in upscale_stream(eng, decoded(), batch=batch, tile=tile, halo=halo):
            mpix += out.shape[0] * out.shape[1] / 1e6
            saves.append(pool.submit(encode, name, out))
            while len(saves) > ahead:
                written.append(saves.pop(0).result())
        written += [f.result() for f in saves]
    sec = time.perf_counter() - t0
    log(f"[INFO] {len(written)} images, {mpix:.1f} output Mpix in {sec * 1e3:.1f} ms "
        f"({sec * 1e3 / max(len(written), 1):.2f} ms/image, {len(written) / sec:.2f} FPS, {mpix / sec:.1f} Mpix/s incl. codecs)")
    return written, mpix, sec


def _worker(rank, world, args):
    import torch
    torch.cuda.set_device(rank)
    names = list_images(args.input_dir)
    mine = [names[i] for i in sharding.shard_range(len(names), rank, world)]
    model = load_model(args.model, compute_dtype=args.dtype, device=rank)
    run_directory(model, args.input_dir, args.output_dir, mine, args.batch, tuple(args.tile) if args.tile else None,
                  args.halo, args.suffix, log=lambda m: print(f"[GPU {rank}] {m}", flush=True))


def main(argv=None):
    ap = argparse.ArgumentParser(prog="superresolution_amd.batch", description=__doc__.split("\n\n")[0])
    ap.add_argument("model")
    ap.add_argument("input_dir")
    ap.add_argument("output_dir")
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--dtype", default=None, choices=["f16", "bf16", "f32"])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--tile", type=int, nargs=2, metavar=("H", "W"), default=None)
    ap.add_argument("--halo", type=int, default=16)
    ap.add_argument("--suffix", default="")
    args = ap.parse_args(argv)
    if not os.path.isdir(args.input_dir):
        print(f"[ERROR] not a directory: {args.input_dir}", file=sys.stderr)
        return -1
    if not os.path.exists(args.model):
        print(f"[ERROR] model file not found: {args.model}", file=sys.stderr)
        return -1
    if args.gpus > 1:
        import torch.multiprocessing as mp
        mp.spawn(_worker, args=(args.gpus, args), nprocs=args.gpus, join=True)
    else:
        _worker(0, 1, args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
