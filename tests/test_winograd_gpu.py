"""The Winograd F(2x2,3x3) form of the 3x3 convolution (srmi_conv_desc_t::algo = SRMI_ALGO_WINOGRAD, csrc/wino3x3.hip),
called through the C ABI.

 * against a CPU emulation with the kernel's own rounding schedule (transformed weights rounded once to f16, every add
   of the input transform rounded to f16, f32 accumulation and output transform; tools/winograd_emulation.py): what is
   left is f32 summation order;
 * against the exact fp32 convolution (nn.Conv2d, the reference's arithmetic) at the f16 tolerance the extra roundings
   of the transform allow;
 * the fused epilogue forms the dense blocks use (realesrgan_model.py:25-32,49): LeakyReLU into a channel slice with
   8-bit rounding, two residuals with a second output, half and single precision residual types;
 * ragged sizes (odd heights and widths, tiles cut by both edges), batches, and writes confined to the output slice;
 * the whole network with the Winograd form on and off against the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sr_oracle as O
from superresolution_amd import _lib, ops, synth
from tests import helpers as H
from tools import winograd_emulation as W

pytestmark = pytest.mark.gpu


def rnd(key, shape, seed=0, amp=1.0):
    return torch.from_numpy(synth.hash_uniform(key, shape, seed)) * amp


def h16(x):
    return x.to(torch.float16).to(torch.float32)


ALGOS = [(_lib.ALGO_WINOGRAD, "F(2x2,3x3)"), (_lib.ALGO_WINOGRAD_X, "F(2,3) along x")]


def emulated(x, wt, b, algo=_lib.ALGO_WINOGRAD):
    if algo == _lib.ALGO_WINOGRAD_X:
        return W.wino1d_conv(x, W.wino1d_weights(wt, W.r16), b, W.r16)
    U = W.wino_weights(wt, W.r16)
    return W.wino_conv(x, U, b, W.r16)


def close(got, ref, rel, what):
    scale = max(float(ref.abs().max()), 1e-6)
    err = float((got - ref).abs().max())
    assert err <= rel * scale, f"{what}: max err {err:.3e} > {rel:.1e} * {scale:.3e}"


@pytest.mark.parametrize("algo,aname", ALGOS)
@pytest.mark.parametrize("cin,cout,n,h,w", [(64, 32, 1, 32, 64), (96, 32, 2, 37, 53), (192, 64, 1, 21, 70), (128, 32, 1, 1, 1),
                                              (160, 32, 1, 17, 33), (64, 64, 3, 16, 31), (64, 32, 1, 9, 131)])
def test_winograd_conv_against_its_emulation_and_the_exact_convolution(cin, cout, n, h, w, algo, aname):
    x = h16(rnd("x", (n, cin, h, w), 1))
    wt = rnd("w", (cout, cin, 3, 3), 2, 0.1)
    b = rnd("b", (cout,), 3)
    pc = ops.PackedConv(wt.numpy(), b.numpy(), _lib.F16, winograd=algo)
    got = ops.conv3x3(ops.Planes.from_nchw(x, _lib.F16), pc, out_dtype=_lib.F32).to_nchw().cpu()
    emu = emulated(x, wt, b, algo)
    exact = F.conv2d(x, wt, b, padding=1)
    close(got, emu, 2e-5, f"{aname} {cin}->{cout} {n}x{h}x{w} against its emulation")
    # the transform's roundings: a handful of 2^-11 steps on sums of four activations, and on the weights
    close(got, exact, 4e-3, f"winograd {cin}->{cout} {n}x{h}x{w} against the exact convolution")
    direct = ops.conv3x3(ops.Planes.from_nchw(x, _lib.F16), ops.PackedConv(wt.numpy(), b.numpy(), _lib.F16, cout_tile=32),
                         out_dtype=_lib.F32).to_nchw().cpu()
    print(f"{aname} {cin}->{cout} {n}x{h}x{w}: |winograd - exact| {float((got - exact).abs().max()):.3e}, |direct - exact| "
          f"{float((direct - exact).abs().max()):.3e} on values up to {float(exact.abs().max()):.3g}")


@pytest.mark.parametrize("algo,aname", ALGOS)
def test_winograd_dense_block_epilogues(algo, aname):
    """conv1-4 form: LeakyReLU, 8-significant-bit rounding, stored into a channel slice of the dense buffer (nothing else
    touched); conv5 form: x5 * 0.2 + x, then * 0.2 + R with a second output, residuals in half and in single precision."""
    n, h, w = 2, 37, 75
    x = h16(rnd("x", (n, 96, h, w), 1))
    wt = rnd("w", (32, 96, 3, 3), 2, 0.1)
    b = rnd("b", (32,), 3)
    pc = ops.PackedConv(wt.numpy(), b.numpy(), _lib.F16, winograd=algo)
    dense = ops.Planes(n, 192, h, w, _lib.F16)
    dense.buf.fill_(7.0)
    ops.conv3x3(ops.Planes.from_nchw(x, _lib.F16), pc, act=_lib.ACT_LRELU, out=dense, out_c0=96, out_round8=True)
    got = dense.to_nchw().cpu()
    ref = F.leaky_relu(emulated(x, wt, b, algo), 0.2).to(torch.bfloat16).to(torch.float32)
    # a value on a rounding boundary may fall to the other side through f32 summation order: one 8-bit step
    close(got[:, 96:128], ref, 2.0 ** -7, "slice store with 8-bit rounding")
    assert (got[:, 96:128] == got[:, 96:128].to(torch.bfloat16).to(torch.float32)).all()
    assert (got[:, :96] == 7.0).all() and (got[:, 128:] == 7.0).all(), "wrote outside its channel slice"
    # conv5
    x5 = h16(rnd("x5", (n, 192, h, w), 4))
    w5 = rnd("w5", (64, 192, 3, 3), 5, 0.05)
    b5 = rnd("b5", (64,), 6)
    pc5 = ops.PackedConv(w5.numpy(), b5.numpy(), _lib.F16, winograd=algo)
    r1, r2 = rnd("r1", (n, 64, h, w), 7), rnd("r2", (n, 64, h, w), 8)
    e5 = emulated(x5, w5, b5, algo)
    for rdt in (_lib.F16, _lib.F32):
        qr = h16 if rdt == _lib.F16 else (lambda t: t)
        ref = (e5 * 0.2 + qr(r1)) * 0.2 + qr(r2)
        out2 = ops.Planes(n, 64, h, w, rdt)
        out = ops.conv3x3(ops.Planes.from_nchw(x5, _lib.F16), pc5, alpha=0.2, res1=ops.Planes.from_nchw(r1, rdt), beta=0.2,
                          res2=ops.Planes.from_nchw(r2, rdt), out2=out2)
        close(out2.to_nchw().cpu(), ref, 2.0 ** -10 if rdt == _lib.F16 else 2e-5, f"conv5 second output, residual type {rdt}")
        close(out.to_nchw().cpu(), ref, 2.0 ** -10, f"conv5 output, residual type {rdt}")
        # one residual, no second output (rdb1 / rdb2 with a half-precision trunk)
        one = ops.conv3x3(ops.Planes.from_nchw(x5, _lib.F16), pc5, alpha=0.2, res1=ops.Planes.from_nchw(r1, rdt),
                          out_dtype=_lib.F32 if rdt == _lib.F32 else None)
        close(one.to_nchw().cpu(), e5 * 0.2 + qr(r1), 2.0 ** -10, f"conv5 with one residual, type {rdt}")


def test_winograd_rejects_what_it_does_not_cover():
    L = _lib.lib()
    x = ops.Planes(1, 48, 16, 16, _lib.F16, zero=True)
    with pytest.raises(_lib.SrmiError, match="multiple of 32"):
        ops.conv3x3(x, ops.PackedConv(np.zeros((32, 48, 3, 3), np.float32), None, _lib.F16, winograd=True))
    x = ops.Planes(1, 64, 16, 16, _lib.F16, zero=True)
    pc = ops.PackedConv(np.zeros((32, 64, 3, 3), np.float32), None, _lib.F16, winograd=True)
    with pytest.raises(_lib.SrmiError, match="Winograd"):
        ops.conv3x3(x, pc, up2=True)
    with pytest.raises(_lib.SrmiError):
        ops.PackedConv(np.zeros((32, 64, 3, 3), np.float32), None, _lib.BF16, winograd=True)
    assert L.srmi_pack_wino3x3_bytes(32, 40, None) == 0


@pytest.mark.parametrize("trunk_f32", [False, True])
def test_rrdbnet_with_and_without_the_winograd_form(trunk_f32):
    """Three blocks at a ragged size, u8 in -> u8 out: both forms within the half-precision bound of the oracle; the
    switch takes effect on the next call and back."""
    meta = dict(arch="rrdbnet", num_block=3, scale=4, seed=61, input="unit")
    model, sd = H.build_model(meta, "f16", trunk_f32=trunk_f32)
    eng = model.engine()
    default = eng.winograd()
    imgs = np.stack([synth.smooth_image(45, 67, 62), synth.random_image(45, 67, 63)])
    with torch.no_grad():
        ref = O.post_unit(H.oracle_forward(meta, sd, O.pre_unit(imgs)), O.ROUND_HALF_UP)
    outs = {}
    for on in (1, 0, 2, 1):
        eng.set_winograd(on)
        got = model.forward_u8(imgs, _lib.ROUND_HALF_UP).numpy()
        d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
        print(f"winograd={on} trunk_f32={trunk_f32}: max u8 diff {d.max()}, PSNR {O.psnr_u8(got, ref):.2f} dB")
        assert d.max() <= 1 and O.psnr_u8(got, ref) >= 50.0
        if on in outs:
            assert np.array_equal(outs[on], got)
        outs[on] = got
    eng.set_winograd(default)
    assert not np.array_equal(outs[1], outs[0]) and not np.array_equal(outs[2], outs[0]), "the switch had no effect"
