"""srmi_rcab_fused (csrc/rcab_fused.hip): both convolutions of a residual channel-attention block (RCAB.forward,
rcan_model.py:79-97) in one kernel, the tensor between them kept on the chip.

 * against the two-launch form through the same C ABI (srmi_conv3x3 with a gated input, then srmi_conv3x3 with the
   attention tail): the block body and the stored block input bit for bit (same order of multiply-accumulates), the
   attention scale to f32 rounding (the pooled sums see other tile shapes);
 * against torch's fp32 ops on the same f16 operands at the f16 tolerance;
 * ragged sizes (tiles cut by both edges, frames smaller than a tile), batches, a plain (ungated) input;
 * the whole network with the one-kernel blocks on and off against the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sr_oracle as O
from superresolution_amd import _lib, ops, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


def rnd(key, shape, seed=0, amp=1.0):
    return torch.from_numpy(synth.hash_uniform(key, shape, seed)) * amp


def h16(x):
    return x.to(torch.float16).to(torch.float32)


def _block(seed, n, h, w, c=64, cr=4):
    x = h16(rnd("x", (n, c, h, w), seed))
    yp = h16(rnd("yp", (n, c, h, w), seed + 1))
    g = (rnd("g", (n, c), seed + 2) * 0.5 + 0.5).cuda()
    w1, b1 = h16(rnd("w1", (c, c, 3, 3), seed + 3, 0.08)), rnd("b1", (c,), seed + 4)
    w2, b2 = h16(rnd("w2", (c, c, 3, 3), seed + 5, 0.08)), rnd("b2", (c,), seed + 6)
    att = (rnd("a1", (cr, c), seed + 7, 0.3).numpy(), rnd("ab1", (cr,), seed + 8).numpy(),
           rnd("a2", (c, cr), seed + 9, 0.3).numpy(), rnd("ab2", (c,), seed + 10).numpy())
    return x, yp, g, (w1, b1), (w2, b2), att


@pytest.mark.parametrize("n,h,w", [(1, 8, 32), (2, 37, 53), (1, 5, 7), (1, 16, 64), (1, 41, 97)])
@pytest.mark.parametrize("gated", [True, False])
def test_one_kernel_block_equals_the_two_launch_form(n, h, w, gated):
    x, yp, g, (w1, b1), (w2, b2), att = _block(11, n, h, w)
    pc1 = ops.PackedConv(w1.numpy(), b1.numpy(), _lib.F16, cout_tile=64)
    pc2 = ops.PackedConv(w2.numpy(), b2.numpy(), _lib.F16, cout_tile=64)
    X, Y = ops.Planes.from_nchw(x, _lib.F16), ops.Planes.from_nchw(yp, _lib.F16)
    alpha = 0.7
    if gated:
        s_two, s_one = ops.Planes(n, 64, h, w, _lib.F16, zero=True), ops.Planes(n, 64, h, w, _lib.F16, zero=True)
        t = ops.conv3x3(X, pc1, act=_lib.ACT_RELU, rows_per_wave=2, in_gate=Y, gate=g, gate_alpha=alpha, in_sum=s_two)
        one, scale_one = ops.rcab_fused(X, pc1, pc2, in_gate=Y, gate=g, gate_alpha=alpha, in_sum=s_one, attention=att)
        assert torch.equal(s_one.to_nchw().cpu(), s_two.to_nchw().cpu()), "stored block input differs"
        # (tv * g) * alpha + x in f32 (the kernels may contract the last two into one fused multiply-add), rounded to f16
        xin = s_one.to_nchw().cpu()
        want = yp * g.cpu()[:, :, None, None] * alpha + x
        assert float((xin - want).abs().max()) <= 2.0 ** -10 * float(want.abs().max())
    else:
        t = ops.conv3x3(X, pc1, act=_lib.ACT_RELU, rows_per_wave=2)
        one, scale_one = ops.rcab_fused(X, pc1, pc2, attention=att)
        xin = x
    two, scale_two = ops.conv3x3(t, pc2, rows_per_wave=2, attention=att)
    a, b = one.to_nchw().cpu(), two.to_nchw().cpu()
    assert torch.equal(a, b), f"block body differs: max {float((a - b).abs().max()):.3e}"
    assert float((scale_one - scale_two).abs().max()) <= 2e-6
    # and the reference's arithmetic on the same operands
    tt = h16(F.relu(F.conv2d(xin, w1, b1, padding=1)))
    ref = F.conv2d(tt, w2, b2, padding=1)
    err = float((a - ref).abs().max())
    assert err <= 2.0 ** -9 * max(float(ref.abs().max()), 1e-6), err
    mean = ref.mean(dim=(2, 3))
    z = F.relu(mean @ torch.from_numpy(att[0]).T + torch.from_numpy(att[1]))
    sref = torch.sigmoid(z @ torch.from_numpy(att[2]).T + torch.from_numpy(att[3]))
    assert float((scale_one.cpu() - sref).abs().max()) <= 2e-3


@pytest.mark.parametrize("n,h,w", [(1, 8, 32), (2, 37, 53), (1, 5, 7), (1, 41, 97)])
def test_one_kernel_block_with_an_f32_trunk_equals_the_two_launch_form(n, h, w):
    """`in` and in_sum as f32 tensors (the residual trunk kept in single precision): the stored block input (f32) and
    the block body bit for bit those of srmi_conv3x3's gated f32 input followed by the second convolution."""
    x, yp, g, (w1, b1), (w2, b2), att = _block(21, n, h, w)
    x = x + rnd("xlo", (n, 64, h, w), 77, 2.0 ** -13)          # bits below f16 precision: the trunk really is f32
    pc1 = ops.PackedConv(w1.numpy(), b1.numpy(), _lib.F16, cout_tile=64)
    pc2 = ops.PackedConv(w2.numpy(), b2.numpy(), _lib.F16, cout_tile=64)
    X, Y = ops.Planes.from_nchw(x, _lib.F32), ops.Planes.from_nchw(yp, _lib.F16)
    alpha = 0.7
    s_two, s_one = ops.Planes(n, 64, h, w, _lib.F32, zero=True), ops.Planes(n, 64, h, w, _lib.F32, zero=True)
    t = ops.conv3x3(X, pc1, act=_lib.ACT_RELU, rows_per_wave=2, in_gate=Y, gate=g, gate_alpha=alpha, in_sum=s_two)
    one, scale_one = ops.rcab_fused(X, pc1, pc2, in_gate=Y, gate=g, gate_alpha=alpha, in_sum=s_one, attention=att)
    assert one.dtype == _lib.F16
    assert torch.equal(s_one.to_nchw().cpu(), s_two.to_nchw().cpu()), "stored block input differs"
    want = yp * g.cpu()[:, :, None, None] * alpha + x
    assert float((s_one.to_nchw().cpu() - want).abs().max()) <= 2.0 ** -22 * float(want.abs().max())
    two, scale_two = ops.conv3x3(t, pc2, rows_per_wave=2, attention=att)
    a, b = one.to_nchw().cpu(), two.to_nchw().cpu()
    assert torch.equal(a, b), f"block body differs: max {float((a - b).abs().max()):.3e}"
    assert float((scale_one - scale_two).abs().max()) <= 2e-6
    # an f32 `in` needs the gated form (a plain input is the trunk's f16 copy)
    with pytest.raises(_lib.SrmiError):
        ops.rcab_fused(X, pc1, pc2)


def test_one_kernel_block_rejects_what_it_does_not_cover():
    x = ops.Planes(1, 128, 16, 16, _lib.F16, zero=True)
    pc = ops.PackedConv(np.zeros((128, 128, 3, 3), np.float32), None, _lib.F16, cout_tile=64)
    with pytest.raises(_lib.SrmiError, match="64 channels"):
        ops.rcab_fused(x, pc, pc)
    x = ops.Planes(1, 64, 16, 16, _lib.BF16, zero=True)
    pc = ops.PackedConv(np.zeros((64, 64, 3, 3), np.float32), None, _lib.BF16, cout_tile=64)
    with pytest.raises(_lib.SrmiError, match="f16"):
        ops.rcab_fused(x, pc, pc)


def test_rcan_with_one_kernel_blocks_against_the_oracle():
    """Two groups of three blocks at a ragged size, f16 trunk (the benchmark's RCAN configuration): the one-kernel blocks
    (default) and the two-launch form both within the half-precision bound of the oracle; batches through the lanes give
    the single-frame bytes."""
    meta = dict(arch="rcan", scale=2, n_resgroups=2, n_resblocks=3, seed=91, input="range")
    model, sd = H.build_model(meta, "f16", trunk_f32=False)
    eng = model.engine()
    assert eng.fused_rcab()
    imgs = np.stack([synth.smooth_image(45, 67, 92), synth.random_image(45, 67, 93)])
    with torch.no_grad():
        ref = O.post_range(H.oracle_forward(meta, sd, O.pre_raw(imgs)), O.ROUND_HALF_UP)
    outs = {}
    for on in (True, False):
        eng.set_fused_rcab(on)
        got = model.forward_u8(imgs, _lib.ROUND_HALF_UP).numpy()
        d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
        print(f"one-kernel blocks={on}: max u8 diff {d.max()}, PSNR {O.psnr_u8(got, ref):.2f} dB")
        assert d.max() <= 1 and O.psnr_u8(got, ref) >= 50.0
        outs[on] = got
    eng.set_fused_rcab(True)
    batch = torch.from_numpy(np.concatenate([imgs] * 3)).cuda()
    many = eng.forward_u8(batch, _lib.ROUND_HALF_UP).cpu().numpy()
    for i in range(6):
        assert np.array_equal(many[i], outs[True][i % 2]), i


def test_rcan_with_an_f32_trunk_runs_one_kernel_blocks():
    """The modules' and runners' default (f16 operands, f32 residual trunk) also runs its blocks as one kernel each:
    within the bound of the oracle, and next to the two-launch plan's bytes (block bodies are bit-identical; the pooled
    means are summed over other tile shapes, so an attention scale may differ in its last f32 bit and a byte of the
    output with it)."""
    meta = dict(arch="rcan", scale=2, n_resgroups=2, n_resblocks=3, seed=95, input="range")
    model, sd = H.build_model(meta, "f16", trunk_f32=True)
    eng = model.engine()
    assert eng.fused_rcab()
    imgs = np.stack([synth.smooth_image(45, 67, 96), synth.random_image(45, 67, 97)])
    with torch.no_grad():
        ref = O.post_range(H.oracle_forward(meta, sd, O.pre_raw(imgs)), O.ROUND_HALF_UP)
    one = model.forward_u8(imgs, _lib.ROUND_HALF_UP).numpy()
    eng.set_fused_rcab(False)
    two = model.forward_u8(imgs, _lib.ROUND_HALF_UP).numpy()
    d12 = np.abs(one.astype(np.int32) - two.astype(np.int32))
    # (measured: 0.18 % of the bytes differ, by one: scales that differ by 1e-6 move values of order 100 by 1e-4 per block)
    assert d12.max() <= 1 and (d12 != 0).mean() < 1e-2, (d12.max(), (d12 != 0).mean())
    for got in (one, two):
        d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
        assert d.max() <= 1 and O.psnr_u8(got, ref) >= 50.0


@pytest.mark.parametrize("trunk_f32", [False, True])
def test_edsr_with_64_features_runs_its_resblocks_as_one_kernel_each(trunk_f32):
    """The EDSR-baseline family (64 features; here 16 blocks, x4; the synthetic weights' gains are made for the
    reference's default res_scale 0.1): conv -> ReLU -> conv of a ResBlock (edsr_model.py:44-47) in one launch, the
    residual add left to the consumer's gated input with a gate vector of ones.  Within the half-precision bound of the
    oracle like the two-launch plan, and next to its bytes (the block output is formed from the f16-rounded body instead
    of the f32 accumulator)."""
    meta = dict(arch="edsr", n_resblocks=16, n_feats=64, scale=4, seed=97, input="range")
    model, sd = H.build_model(meta, "f16", trunk_f32=trunk_f32)
    eng = model.engine()
    assert eng.fused_rcab()
    imgs = np.stack([synth.smooth_image(61, 83, 98), synth.random_image(61, 83, 99)])
    with torch.no_grad():
        ref = O.post_range(H.oracle_forward(meta, sd, O.pre_raw(imgs)), O.ROUND_HALF_UP)
    outs = {}
    for on in (True, False):
        eng.set_fused_rcab(on)
        got = model.forward_u8(imgs, _lib.ROUND_HALF_UP).numpy()
        d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
        print(f"edsr 16x64 trunk_f32={trunk_f32} one-kernel blocks={on}: max u8 diff {d.max()}, PSNR {O.psnr_u8(got, ref):.2f} dB")
        assert d.max() <= 1 and O.psnr_u8(got, ref) >= 50.0
        outs[on] = got
    d12 = np.abs(outs[True].astype(np.int32) - outs[False].astype(np.int32))
    assert d12.max() <= 1
    eng.set_fused_rcab(True)
    batch = torch.from_numpy(np.concatenate([imgs] * 3)).cuda()
    many = eng.forward_u8(batch, _lib.ROUND_HALF_UP).cpu().numpy()
    for i in range(6):
        assert np.array_equal(many[i], outs[True][i % 2]), i
