"""Known-answer test on the OFFICIAL checkpoints: the reference's three committed output PNGs
(tests/golden/ref_png/, made by its test_pytorch.py scripts with EDSR_x4.pt, RCAN_BIX4.pt and
RealESRGAN_x4plus.pth) against this engine's u8 path, with the rounding mode each script used.

The checkpoints cannot be fetched here: the test runs only when $SRMI_CKPT_DIR holds them and skips otherwise.
It is the one test that measures the half-precision margin (and RRDBNet's two-precision dense blocks,
growth_round8) on real weights rather than on synthetic ones.  Files are opened with
torch.load(weights_only=True) by the package's loaders."""
import os

import numpy as np
import pytest
import torch

from oracle import sr_oracle as O
from superresolution_amd import _lib

pytestmark = pytest.mark.gpu

PNG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_png")
CKPT = os.environ.get("SRMI_CKPT_DIR", "")

CASES = {
    "edsr": ("EDSR_x4.pt", "edsr_input_text_256x256.png", "edsr_pytorch_256x256.png", _lib.ROUND_TRUNC),
    "rcan": ("RCAN_BIX4.pt", "input_510x339.png", "rcan_pytorch_510x339.png", _lib.ROUND_TRUNC),
    "realesrgan": ("RealESRGAN_x4plus.pth", "input_510x339.png", "realesrgan_test_rknn_pytorch.png", _lib.ROUND_HALF_EVEN),
}


def _load(arch, path, **kw):
    from superresolution_amd import load_edsr_from_checkpoint, load_rcan_from_checkpoint, load_realesrgan_from_checkpoint
    if arch == "edsr":
        return load_edsr_from_checkpoint(path, 4, **kw)
    if arch == "rcan":
        return load_rcan_from_checkpoint(path, 4, **kw)
    return load_realesrgan_from_checkpoint(path, **kw)[0]


def _variants(arch):
    out = [("f32", {}), ("f16", {})]
    if arch == "realesrgan":
        out.append(("f16", {"growth_round8": False}))
    return out


@pytest.mark.parametrize("arch", sorted(CASES))
def test_official_checkpoint_reproduces_the_reference_png(arch):
    from PIL import Image
    ckpt, inp, want_png, mode = CASES[arch]
    path = os.path.join(CKPT, ckpt)
    if not CKPT or not os.path.exists(path):
        pytest.skip(f"official checkpoint {ckpt} not supplied (set SRMI_CKPT_DIR)")
    img = np.asarray(Image.open(os.path.join(PNG, inp)).convert("RGB"))
    want = np.asarray(Image.open(os.path.join(PNG, want_png)).convert("RGB"))
    for dtype, kw in _variants(arch):
        model = _load(arch, path, compute_dtype=dtype, **kw)
        got = model.forward_u8(img, mode).numpy()
        assert got.shape == want.shape
        diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
        psnr = O.psnr_u8(got, want)
        print(f"KAT {arch} {dtype} {kw}: max u8 diff {diff.max()}, differing {100 * (diff != 0).mean():.3f} %, PSNR {psnr:.2f} dB")
        assert diff.max() <= 1 and psnr >= 50.0, (arch, dtype, kw)


def test_kat_fixtures_are_readable_and_consistent():
    """Runs everywhere: the fixture images exist and each golden output is exactly 4x its input."""
    from PIL import Image
    for arch, (_, inp, want_png, _) in CASES.items():
        a, b = Image.open(os.path.join(PNG, inp)), Image.open(os.path.join(PNG, want_png))
        assert (b.size[0], b.size[1]) == (4 * a.size[0], 4 * a.size[1]), arch
