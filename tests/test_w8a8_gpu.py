"""The int8 (w8a8) path: the reference's low-precision deployment (realesrgan/rknn/python/convert.py:198-278,
quantized_dtype w8a8 + calibration images in [0,1]; its own result: 33.34 dB against its FP16 model,
realesrgan/rknn/MIGRATION_REPORT.md:368-376) on the MI355X int8 matrix cores.

 * operator level: int8 x int8 -> int32 is exact, so the convolution must equal an integer reference exactly;
 * model level: RRDBNet with calibrated int8 dense blocks against the fp32 oracle, PSNR recorded and bounded."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sr_oracle as O
from superresolution_amd import _lib, ops, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _ints(key, shape, seed, lo=-127, hi=127):
    u = synth.hash_uniform(key, shape, seed)                 # [-1, 1)
    return np.clip(np.rint(u * hi), lo, hi).astype(np.float32)


@pytest.mark.parametrize("cin,cout,mb", [(64, 32, 3), (96, 32, 2), (192, 64, 3)])
def test_int8_conv_is_exact(cin, cout, mb):
    n, h, w = 2, 29, 45
    xq = torch.from_numpy(_ints("x", (n, cin, h, w), 1))
    wq = torch.from_numpy(_ints("w", (cout, cin, 3, 3), 2))
    ws = (synth.hash_uniform("ws", (cout,), 3) * 0.5 + 1.0).astype(np.float32) * 1e-4
    b = synth.hash_uniform("b", (cout,), 4).astype(np.float32)
    acc = F.conv2d(xq.double(), wq.double(), None, padding=1)              # exact integers
    assert acc.abs().max() < 2 ** 31
    ref = acc * torch.from_numpy(ws).double().view(1, -1, 1, 1) + torch.from_numpy(b).double().view(1, -1, 1, 1)
    xp = ops.Planes.from_nchw(xq, _lib.I8, qscale=1.0)
    assert torch.equal(xp.buf.cpu().permute(0, 1, 4, 2, 3).reshape(n, -1, h, w)[:, :cin].float(), xq)
    pc = ops.PackedConv(wq.numpy(), b, _lib.I8, cout_tile=32, wscale=ws)
    got = ops.conv3x3(xp, pc, out_dtype=_lib.F32, rows_per_wave=mb).to_nchw().cpu().double()
    err = (got - ref).abs().max().item()
    print(f"int8 conv {cin}->{cout} mb{mb}: max |err| {err:.3g} on values up to {ref.abs().max():.3g}")
    assert err <= 1e-6 * ref.abs().max().item()
    # LeakyReLU + int8 output with its own scale: at most one quantisation step off (ties), almost never
    qs = 127.0 / float(torch.where(ref > 0, ref, 0.2 * ref).abs().max())
    out = ops.conv3x3(xp, pc, act=_lib.ACT_LRELU, out_dtype=_lib.I8, out_qscale=qs, rows_per_wave=mb)
    q = out.buf.cpu().permute(0, 1, 4, 2, 3).reshape(n, -1, h, w)[:, :cout].double()
    want = torch.clamp(torch.round(torch.where(ref > 0, ref, 0.2 * ref) * qs), -127, 127)
    d = (q - want).abs()
    assert d.max() <= 1 and (d != 0).double().mean() < 1e-3


def test_int8_conv_block_tail_form():
    """The last conv of a dense block in w8a8 mode: int8 in, two half-precision residuals, the result written
    quantised (next conv input) and in f16 (next residual) (realesrgan_model.py:30-32,49)."""
    n, h, w, cin, cout = 1, 25, 40, 192, 64
    xq = torch.from_numpy(_ints("x", (n, cin, h, w), 5))
    wq = torch.from_numpy(_ints("w", (cout, cin, 3, 3), 6))
    ws = np.full(cout, 3e-5, np.float32)
    b = synth.hash_uniform("b", (cout,), 7).astype(np.float32)
    r1 = torch.from_numpy(synth.hash_uniform("r1", (n, cout, h, w), 8)).half().float()
    r2 = torch.from_numpy(synth.hash_uniform("r2", (n, cout, h, w), 9)).half().float()
    acc = F.conv2d(xq.double(), wq.double(), None, padding=1)
    v = acc * torch.from_numpy(ws).double().view(1, -1, 1, 1) + torch.from_numpy(b).double().view(1, -1, 1, 1)
    ref = (v * 0.2 + r1.double()) * 0.2 + r2.double()
    xp = ops.Planes.from_nchw(xq, _lib.I8, qscale=1.0)
    pc = ops.PackedConv(wq.numpy(), b, _lib.I8, cout_tile=32, wscale=ws)
    qs = 127.0 / float(ref.abs().max())
    out2 = ops.Planes(n, cout, h, w, _lib.F16)
    out = ops.conv3x3(xp, pc, alpha=0.2, res1=ops.Planes.from_nchw(r1, _lib.F16), beta=0.2,
                      res2=ops.Planes.from_nchw(r2, _lib.F16), out_dtype=_lib.I8, out_qscale=qs, out2=out2)
    assert (out2.to_nchw().cpu().double() - ref).abs().max() <= 2e-3 * ref.abs().max()
    q = out.buf.cpu().permute(0, 1, 4, 2, 3).reshape(n, -1, h, w)[:, :cout].double()
    d = (q - torch.clamp(torch.round(ref * qs), -127, 127)).abs()
    assert d.max() <= 1 and (d != 0).double().mean() < 2e-2


def _calibration_frames(h, w, n=6):
    # smooth natural-like frames plus noise frames, as the reference's synthetic calibration set mixes gradients,
    # patterns and noise (edsr/rknn/python/prepare_quantization_dataset.py:72-115)
    return np.stack([synth.smooth_image(h, w, 900 + i) for i in range(n - 2)] +
                    [synth.random_image(h, w, 950 + i) for i in range(2)])


@pytest.mark.parametrize("num_block", [2, 23])
def test_rrdbnet_w8a8_against_the_oracle(num_block):
    meta = dict(arch="rrdbnet", num_block=num_block, scale=4, seed=81, input="unit")
    model, sd = H.build_model(meta, "f16")
    h, w = 48, 64
    img = synth.smooth_image(h, w, 82)
    f16_out = model.forward_u8(img, _lib.ROUND_HALF_UP).numpy()
    stats = model.quantize_w8a8(_calibration_frames(h, w))
    assert stats.shape == (5 * (3 * num_block + 1),) and (stats[:-4] > 0).all()
    got = model.forward_u8(img, _lib.ROUND_HALF_UP).numpy()
    with torch.no_grad():
        ref = O.post_unit(H.oracle_forward(meta, sd, O.pre_unit(img)), O.ROUND_HALF_UP)[0]
    diff = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    psnr = O.psnr_u8(got, ref)
    print(f"rrdbnet b{num_block} w8a8: PSNR {psnr:.2f} dB vs the fp32 oracle, max u8 diff {diff.max()}, "
          f"differing {100 * (diff != 0).mean():.1f} % (f16: {O.psnr_u8(f16_out, ref):.2f} dB)")
    # the reference's own INT8 deployment reaches 33.34 dB against its FP16 model (MIGRATION_REPORT.md:368-376)
    assert psnr >= 33.34
    assert not np.array_equal(got, f16_out)
    # a batch through the lanes gives the same frames; switching the mode off restores the f16 bytes exactly
    outs = model.engine().forward_u8(torch.from_numpy(np.stack([img] * 4)).cuda(), _lib.ROUND_HALF_UP).cpu().numpy()
    assert all(np.array_equal(o, got) for o in outs)
    model.engine().set_w8a8(False)
    assert np.array_equal(model.forward_u8(img, _lib.ROUND_HALF_UP).numpy(), f16_out)


def test_rrdbnet_w8a8_with_heavy_tailed_weights():
    """Per-channel int8 weight scales are set by each channel's largest weight, so a heavy-tailed weight
    distribution (sign(w) |w|^3 at the same RMS: a few weights carry most of the energy) is the hard case for the
    quantiser.  Recorded against the fp32 oracle at full depth; the asserted bar is the reference's own INT8 result."""
    from superresolution_amd import RRDBNet
    meta = dict(arch="rrdbnet", num_block=23, scale=4, seed=83, input="unit")
    sd = H.state_dict_for(meta)
    for k in list(sd):
        if ".rdb" in k and k.endswith(".weight"):
            w = sd[k].astype(np.float64)
            v = np.sign(w) * np.abs(w / np.abs(w).max()) ** 3
            sd[k] = (v * (np.sqrt(np.mean(w * w)) / np.sqrt(np.mean(v * v)))).astype(np.float32)
    h, w_ = 48, 64
    img = synth.smooth_image(h, w_, 84)
    model = RRDBNet(num_block=23, scale=4, compute_dtype="f16")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.quantize_w8a8(_calibration_frames(h, w_))
    got = model.forward_u8(img, _lib.ROUND_HALF_UP).numpy()
    with torch.no_grad():
        ref = O.post_unit(H.oracle_forward(meta, sd, O.pre_unit(img)), O.ROUND_HALF_UP)[0]
    diff = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    psnr = O.psnr_u8(got, ref)
    print(f"rrdbnet b23 w8a8, heavy-tailed weights: PSNR {psnr:.2f} dB vs the fp32 oracle, max u8 diff {diff.max()}")
    assert psnr >= 33.34


def test_w8a8_needs_calibration_and_an_f16_rrdbnet():
    L = _lib.lib()
    meta = dict(arch="rrdbnet", num_block=1, scale=4, seed=83, input="unit")
    model, _ = H.build_model(meta, "f16")
    eng = model.engine()
    assert L.srmi_model_set_w8a8(eng._h, 1) == -1 and b"calibrated" in L.srmi_last_error()
    assert L.srmi_model_calibrate_end(eng._h) == -1
    assert L.srmi_model_calibrate_begin(eng._h) == 0
    assert L.srmi_model_calibrate_end(eng._h) == -1 and b"no forward pass" in L.srmi_last_error()
    edsr, _ = H.build_model(dict(arch="edsr", n_resblocks=1, n_feats=64, scale=2, seed=84, input="range"), "f16")
    assert L.srmi_model_calibrate_begin(edsr.engine()._h) == -1


def test_w8a8_weight_file_round_trip_and_runner(tmp_path):
    """A calibrated model written by export.save_weights carries its activation ranges: the C runner and
    export.load_weights start in w8a8 mode and give the calibrated model's bytes; SRMI_W8A8=0 gives plain f16."""
    import ctypes as C
    import os
    from superresolution_amd import export
    from tests.test_runner_gpu import RealesrganCtx, _take
    L = _lib.lib()
    meta = dict(arch="rrdbnet", num_block=1, scale=4, seed=85, input="unit")
    model, _ = H.build_model(meta, "f16")
    img = synth.smooth_image(339, 510, 86)
    f16_out = model.forward_u8(img, _lib.ROUND_HALF_UP).numpy()
    model.quantize_w8a8(_calibration_frames(96, 128, 4))
    want = model.forward_u8(img, _lib.ROUND_HALF_UP).numpy()
    assert not np.array_equal(want, f16_out)
    path = str(tmp_path / "RealESRGAN_x4plus_w8a8.srmi")
    export.save_weights(model, path)
    again = export.load_weights(path, compute_dtype="f16")
    assert np.array_equal(again.engine().calibration(), model.engine().calibration())
    assert np.array_equal(again.forward_u8(img, _lib.ROUND_HALF_UP).numpy(), want)
    for env, expect in ((None, want), ("0", f16_out)):
        if env is None:
            os.environ.pop("SRMI_W8A8", None)
        else:
            os.environ["SRMI_W8A8"] = env
        try:
            ctx = RealesrganCtx()
            assert L.realesrgan_init(path.encode(), C.byref(ctx)) == 0, L.srmi_last_error()
            out = C.c_void_p()
            assert L.realesrgan_inference(C.byref(ctx), img.ctypes.data_as(C.c_void_p), C.byref(out)) == 0
            got = _take(out, 1356, 2040)
            assert L.realesrgan_release(C.byref(ctx)) == 0
        finally:
            os.environ.pop("SRMI_W8A8", None)
        assert np.array_equal(got, expect), f"SRMI_W8A8={env}"


def test_weight_file_of_a_model_switched_back_to_f16_comes_back_in_f16(tmp_path):
    """A calibrated model whose int8 form is switched OFF when it is saved (set_w8a8(False), or an experimental
    calibration that was not kept) must not come back in int8: the file then carries no calibration trailer, and
    both readers (export.load_weights, the C runner) give the f16 bytes.  set_calibration(..., w8a8=False) installs
    ranges without switching the mode on."""
    import ctypes as C
    import os
    from superresolution_amd import export
    from tests.test_runner_gpu import RealesrganCtx, _take
    L = _lib.lib()
    meta = dict(arch="rrdbnet", num_block=1, scale=4, seed=87, input="unit")
    model, _ = H.build_model(meta, "f16")
    img = synth.smooth_image(339, 510, 88)
    f16_out = model.forward_u8(img, _lib.ROUND_HALF_UP).numpy()
    stats = model.quantize_w8a8(_calibration_frames(96, 128, 4))
    eng = model.engine()
    assert eng.w8a8()
    q_out = model.forward_u8(img, _lib.ROUND_HALF_UP).numpy()
    assert not np.array_equal(q_out, f16_out)
    eng.set_w8a8(False)
    assert not eng.w8a8()
    path = str(tmp_path / "RealESRGAN_x4plus_f16_again.srmi")
    export.save_weights(model, path)
    with open(path, "rb") as f:
        assert b"SRMIQ001" not in f.read()[-(8 + 4 + 4 * len(stats)):]
    again = export.load_weights(path, compute_dtype="f16")
    assert not again.engine().w8a8() and len(again.engine().calibration()) == 0
    assert np.array_equal(again.forward_u8(img, _lib.ROUND_HALF_UP).numpy(), f16_out)
    os.environ.pop("SRMI_W8A8", None)
    ctx = RealesrganCtx()
    assert L.realesrgan_init(path.encode(), C.byref(ctx)) == 0, L.srmi_last_error()
    out = C.c_void_p()
    assert L.realesrgan_inference(C.byref(ctx), img.ctypes.data_as(C.c_void_p), C.byref(out)) == 0
    got = _take(out, 1356, 2040)
    assert L.realesrgan_release(C.byref(ctx)) == 0
    assert np.array_equal(got, f16_out)
    # ranges installed without switching the mode on: still f16 until set_w8a8(True)
    again.engine().set_calibration(stats, w8a8=False)
    assert not again.engine().w8a8()
    assert np.array_equal(again.forward_u8(img, _lib.ROUND_HALF_UP).numpy(), f16_out)
    again.engine().set_w8a8(True)
    assert np.array_equal(again.forward_u8(img, _lib.ROUND_HALF_UP).numpy(), q_out)


def test_rrdbnet_w8a8_full_size_frame_against_the_oracle():
    """The benchmark's w8a8 configuration (23 blocks, 510x339 -> 2040x1356, calibration on six synthetic frames of
    that size) against the fp32 oracle on one whole frame.  <= 1 LSB is not required of int8; the recorded figures are
    asserted at >= 50 dB (the half-precision bound of the f16 engine) with the maximum LSB difference printed; the
    reference's own INT8 deployment reports 33.34 dB (realesrgan/rknn/MIGRATION_REPORT.md:368-376)."""
    meta = dict(arch="rrdbnet", num_block=23, scale=4, seed=0, input="unit")
    model, sd = H.build_model(meta, "f16", trunk_f32=False)
    h, w = 339, 510
    img = synth.smooth_image(h, w, 7)
    f16_out = model.forward_u8(img, _lib.ROUND_HALF_UP).numpy()
    model.quantize_w8a8(_calibration_frames(h, w), batch=3)
    got = model.forward_u8(img, _lib.ROUND_HALF_UP).numpy()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with torch.no_grad():
        ref = O.post_unit(H.oracle_forward(meta, sd, O.pre_unit(img)), O.ROUND_HALF_UP)[0]
    diff = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    psnr = O.psnr_u8(got, ref)
    print(f"rrdbnet b23 w8a8 at 510x339: PSNR {psnr:.2f} dB vs the fp32 oracle, max u8 diff {diff.max()}, "
          f"> 1 LSB on {100 * (diff > 1).mean():.4f} % of the samples (f16 engine: {O.psnr_u8(f16_out, ref):.2f} dB)")
    assert psnr >= 50.0 and diff.max() <= 3
