"""Operator-level parity on the GPU: every HIP kernel, called through the C ABI, against the CPU
oracle (torch fp32 functional ops, the arithmetic the reference modules use).

Tolerances: f32 kernels 2e-5 relative to the output scale (summation order only differs);
f16/bf16 kernels are fed operands already rounded to the half type, so the remaining difference is
fp32 accumulation order plus ONE output rounding: 2^-10 (f16) / 2^-7 (bf16) relative.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sr_oracle as O
from superresolution_amd import _lib, ops, synth

pytestmark = pytest.mark.gpu

DTYPES = [(_lib.F32, "f32", 2e-5), (_lib.F16, "f16", 2.0 ** -10), (_lib.BF16, "bf16", 2.0 ** -7)]
TORCH_DT = {_lib.F32: torch.float32, _lib.F16: torch.float16, _lib.BF16: torch.bfloat16}


def q(x, dt):
    """Round a float tensor to the compute dtype and back (what the kernel's operands hold)."""
    return x.to(TORCH_DT[dt]).to(torch.float32)


def rnd(key, shape, seed=0, amp=1.0):
    return torch.from_numpy(synth.hash_uniform(key, shape, seed)) * amp


def assert_close(got, ref, rel, what=""):
    scale = max(float(ref.abs().max()), 1e-6)
    err = float((got - ref).abs().max())
    assert err <= rel * scale, f"{what}: max err {err:.3e} > {rel:.1e} * {scale:.3e}"


@pytest.mark.parametrize("dt,name,tol", DTYPES)
@pytest.mark.parametrize("mb", [2, 3])
def test_conv_64_64_relu(dt, name, tol, mb):
    n, h, w = 2, 33, 47
    x = q(rnd("x", (n, 64, h, w), 1), dt)
    wt = q(rnd("w", (64, 64, 3, 3), 2, 0.1), dt)
    b = rnd("b", (64,), 3)
    ref = F.relu(F.conv2d(x, wt, b, padding=1))
    pc = ops.PackedConv(wt.numpy(), b.numpy(), dt)
    out = ops.conv3x3(ops.Planes.from_nchw(x, dt), pc, act=_lib.ACT_RELU, rows_per_wave=mb,
                      out_dtype=_lib.F32)
    assert_close(out.to_nchw().cpu(), ref, 2e-5 if dt == _lib.F32 else 1e-5 * 50, f"conv {name} mb={mb}")
    # and with the output rounded to the compute dtype
    out = ops.conv3x3(ops.Planes.from_nchw(x, dt), pc, act=_lib.ACT_RELU, rows_per_wave=mb)
    assert_close(out.to_nchw().cpu(), ref, tol, f"conv {name} mb={mb} (typed out)")


@pytest.mark.parametrize("dt,name,tol", DTYPES)
@pytest.mark.parametrize("mb", [4])
def test_conv_tall_tiles_32_outputs(dt, name, tol, mb):
    """16-row tiles (32-output workgroups only): ragged bottom and right edges, LeakyReLU, a residual."""
    n, h, w = 2, 53, 75
    x = q(rnd("x", (n, 96, h, w), 1), dt)
    wt = q(rnd("w", (64, 96, 3, 3), 2, 0.1), dt)
    b = rnd("b", (64,), 3)
    r1 = q(rnd("r1", (n, 64, h, w), 4), dt)
    ref = F.leaky_relu(F.conv2d(x, wt, b, padding=1), 0.2) * 0.2 + r1
    pc = ops.PackedConv(wt.numpy(), b.numpy(), dt, cout_tile=32)
    out = ops.conv3x3(ops.Planes.from_nchw(x, dt), pc, act=_lib.ACT_LRELU, rows_per_wave=mb, alpha=0.2,
                      res1=ops.Planes.from_nchw(r1, dt))
    assert_close(out.to_nchw().cpu(), ref, tol, f"conv {name} mb={mb}")
    ref3 = ops.conv3x3(ops.Planes.from_nchw(x, dt), pc, act=_lib.ACT_LRELU, rows_per_wave=3, alpha=0.2,
                       res1=ops.Planes.from_nchw(r1, dt))
    assert torch.equal(out.to_nchw().cpu(), ref3.to_nchw().cpu()), "tile height changes the result"


@pytest.mark.parametrize("dt,name,tol", DTYPES)
def test_rdb_conv5_two_residuals(dt, name, tol):
    # x5*0.2 + x, then *0.2 + R  (realesrgan_model.py:32,49); trunk copies in f32
    n, h, w = 1, 21, 70
    x = q(rnd("x", (n, 192, h, w), 1), dt)
    wt = q(rnd("w", (64, 192, 3, 3), 2, 0.05), dt)
    b = rnd("b", (64,), 3)
    r1 = rnd("r1", (n, 64, h, w), 4)
    r2 = rnd("r2", (n, 64, h, w), 5)
    ref = (F.conv2d(x, wt, b, padding=1) * 0.2 + r1) * 0.2 + r2
    pc = ops.PackedConv(wt.numpy(), b.numpy(), dt)
    out2 = ops.Planes(n, 64, h, w, _lib.F32)
    out = ops.conv3x3(ops.Planes.from_nchw(x, dt), pc, alpha=0.2, res1=ops.Planes.from_nchw(r1, _lib.F32),
                      beta=0.2, res2=ops.Planes.from_nchw(r2, _lib.F32), out2=out2)
    assert_close(out2.to_nchw().cpu(), ref, 2e-5 if dt == _lib.F32 else 5e-4, f"conv5 {name} f32 copy")
    assert_close(out.to_nchw().cpu(), ref, tol, f"conv5 {name}")
    # residual given in the compute dtype
    r1q = q(r1, dt)
    ref = F.conv2d(x, wt, b, padding=1) * 0.2 + r1q
    out = ops.conv3x3(ops.Planes.from_nchw(x, dt), pc, alpha=0.2, res1=ops.Planes.from_nchw(r1q, dt))
    assert_close(out.to_nchw().cpu(), ref, tol, f"conv5 {name} typed residual")


@pytest.mark.parametrize("dt,name,tol", DTYPES)
def test_dense_slice_store_lrelu(dt, name, tol):
    # conv2 of a ResidualDenseBlock: reads channels 0..95 of the 192-channel dense buffer, writes
    # 32 channels at offset 96 (torch.cat replaced by slice stores; realesrgan_model.py:27-28)
    n, h, w = 1, 16, 40
    full = q(rnd("x", (n, 192, h, w), 1), dt)
    wt = q(rnd("w", (32, 96, 3, 3), 2, 0.05), dt)
    b = rnd("b", (32,), 3)
    ref = F.leaky_relu(F.conv2d(full[:, :96], wt, b, padding=1), 0.2)
    dense = ops.Planes.from_nchw(full, dt)
    pc = ops.PackedConv(wt.numpy(), b.numpy(), dt)
    assert pc.cout_tile == 32
    for mb in (2, 3, 4):   # the 32-channel tile supports 2, 3 and 4 rows per wave
        dense = ops.Planes.from_nchw(full, dt)
        ops.conv3x3(dense, pc, act=_lib.ACT_LRELU, out=dense, out_c0=96, rows_per_wave=mb)
        got = dense.to_nchw().cpu()
        assert_close(got[:, 96:128], ref, tol, f"slice {name} mb={mb}")
        assert torch.equal(got[:, :96], full[:, :96]) and torch.equal(got[:, 128:], full[:, 128:])


@pytest.mark.parametrize("dt,name,tol", DTYPES)
def test_nearest_up2_loader(dt, name, tol):
    n, h, w = 1, 13, 19
    x = q(rnd("x", (n, 64, h, w), 1), dt)
    wt = q(rnd("w", (64, 64, 3, 3), 2, 0.1), dt)
    b = rnd("b", (64,), 3)
    ref = F.leaky_relu(F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), wt, b, padding=1), 0.2)
    out = ops.conv3x3(ops.Planes.from_nchw(x, dt), ops.PackedConv(wt.numpy(), b.numpy(), dt),
                      act=_lib.ACT_LRELU, up2=True)
    assert (out.h, out.w) == (2 * h, 2 * w)
    assert_close(out.to_nchw().cpu(), ref, tol, f"up2 {name}")


@pytest.mark.parametrize("dt,name,tol", [(_lib.F16, "f16", 2.0 ** -10), (_lib.BF16, "bf16", 2.0 ** -7)])
@pytest.mark.parametrize("cin,cout,cout_tile,mb", [(64, 64, 32, 3), (64, 64, 64, 2), (96, 32, 32, 2), (32, 128, 64, 3)])
def test_upsampling_conv_subpixel_form(dt, name, tol, cin, cout, cout_tile, mb):
    """nearest-x2 upsampling, then conv (realesrgan_model.py:115-116) as four 2x2 convolutions on the source grid
    (SRMI_ALGO_SUBPIXEL_UP2): against torch on the upsampled tensor, and against the nine-tap kernel reading through the
    x2 address map.  The tap sums are rounded once to the compute type, so the forms agree to a rounding of the weights,
    not bit for bit.  Several tiles both ways, ragged edges, two images."""
    n, h, w = 2, 27, 41
    x = q(rnd("x", (n, cin, h, w), 1), dt)
    wt = q(rnd("w", (cout, cin, 3, 3), 2, 0.1), dt)
    b = rnd("b", (cout,), 3)
    ref = F.leaky_relu(F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), wt, b, padding=1), 0.2)
    xp = ops.Planes.from_nchw(x, dt)
    sub = ops.conv3x3(xp, ops.PackedConv(wt.numpy(), b.numpy(), dt, cout_tile=cout_tile, subpixel_up2=True),
                      act=_lib.ACT_LRELU, up2=True, rows_per_wave=mb)
    assert (sub.c, sub.h, sub.w) == (cout, 2 * h, 2 * w)
    # one more rounding (of the summed taps) than the nine-tap form
    assert_close(sub.to_nchw().cpu(), ref, 2 * tol, f"sub-pixel up2 {name}")
    nine = ops.conv3x3(xp, ops.PackedConv(wt.numpy(), b.numpy(), dt, cout_tile=cout_tile), act=_lib.ACT_LRELU, up2=True,
                       rows_per_wave=mb)
    assert_close(sub.to_nchw().cpu(), nine.to_nchw().cpu(), 2 * tol, f"sub-pixel against nine taps {name}")
    # f32 output: what is left against torch is the rounding of the tap sums and the f32 summation order
    sub32 = ops.conv3x3(xp, ops.PackedConv(wt.numpy(), b.numpy(), dt, cout_tile=cout_tile, subpixel_up2=True),
                        act=_lib.ACT_LRELU, up2=True, rows_per_wave=mb, out_dtype=_lib.F32)
    assert_close(sub32.to_nchw().cpu(), ref, tol, f"sub-pixel up2 {name} (f32 out)")


def test_upsampling_conv_subpixel_form_is_declined_where_it_does_not_apply():
    x = ops.Planes.from_nchw(rnd("x", (1, 64, 8, 8), 1), _lib.F16)
    pc = ops.PackedConv(rnd("w", (64, 64, 3, 3), 2, 0.1).numpy(), None, _lib.F16, subpixel_up2=True)
    with pytest.raises(_lib.SrmiError, match="in_up2"):
        ops.conv3x3(x, pc)                      # not an upsampling convolution
    with pytest.raises(_lib.SrmiError, match="rows_per_wave"):
        ops.conv3x3(x, pc, up2=True, rows_per_wave=4)
    with pytest.raises(_lib.SrmiError):
        ops.PackedConv(rnd("w", (64, 64, 3, 3), 2, 0.1).numpy(), None, _lib.F32, subpixel_up2=True)


@pytest.mark.parametrize("dt,name,tol", DTYPES)
@pytest.mark.parametrize("r", [2, 3])
def test_pixel_shuffle_store(dt, name, tol, r):
    n, h, w = 1, 11, 37
    x = q(rnd("x", (n, 64, h, w), 1), dt)
    wt = q(rnd("w", (64 * r * r, 64, 3, 3), 2, 0.1), dt)
    b = rnd("b", (64 * r * r,), 3)
    ref = F.pixel_shuffle(F.conv2d(x, wt, b, padding=1), r)
    out = ops.conv3x3(ops.Planes.from_nchw(x, dt), ops.PackedConv(wt.numpy(), b.numpy(), dt, ps_r=r))
    assert (out.c, out.h, out.w) == (64, h * r, w * r)
    assert_close(out.to_nchw().cpu(), ref, tol, f"ps{r} {name}")


@pytest.mark.parametrize("dt,name,tol", DTYPES)
def test_rgb_tail(dt, name, tol):
    n, h, w = 2, 19, 45
    x = q(rnd("x", (n, 64, h, w), 1), dt)
    wt = q(rnd("w", (3, 64, 3, 3), 2, 0.1), dt)
    b = rnd("b", (3,), 3) + 0.5
    core = F.conv2d(x, wt, b, padding=1)
    pc = ops.PackedConv(wt.numpy(), b.numpy(), dt, cout_tile=32)
    xp = ops.Planes.from_nchw(x, dt)
    ftol = 2e-5 if dt == _lib.F32 else 5e-4
    # f32 NCHW with a general 3x3 colour matrix (add_mean as a 1x1 conv, edsr_model.py:122)
    mat = np.array([[1.0, 0.1, 0.0], [0.0, 0.9, 0.2], [0.3, 0.0, 1.1]], np.float32)
    tb = np.array([0.5, -0.25, 0.125], np.float32)
    ref = F.conv2d(core, torch.from_numpy(mat).view(3, 3, 1, 1), torch.from_numpy(tb))
    got = ops.conv3x3(xp, pc, tail=dict(mode=_lib.TAIL_F32_NCHW, mat=mat, bias=tb))
    assert_close(got.cpu(), ref, ftol, f"tail f32 {name}")
    # u8 NHWC, each rounding mode (runners' post-processing)
    for mode, omode in [(_lib.ROUND_HALF_EVEN, O.ROUND_HALF_EVEN), (_lib.ROUND_HALF_UP, O.ROUND_HALF_UP),
                        (_lib.ROUND_TRUNC, O.ROUND_TRUNC)]:
        got = ops.conv3x3(xp, pc, tail=dict(mode=_lib.TAIL_U8_NHWC, lo=0.0, hi=1.0, scale=255.0, round=mode))
        ref8 = O.post_unit(core, omode)
        diff = np.abs(got.cpu().numpy().astype(np.int32) - ref8.astype(np.int32))
        assert diff.max() <= 1, f"tail u8 {name} mode {mode}: max diff {diff.max()}"
        assert (diff != 0).mean() < (0.002 if dt == _lib.F32 else 0.1)
        assert 0.02 < (ref8 == 0).mean() < 0.9  # clamping is exercised


@pytest.mark.parametrize("cin,cout_tile", [(64, 64), (128, 32), (192, 32), (256, 64), (32, 32), (80, 32)])
@pytest.mark.parametrize("dt,name", [(_lib.F16, "f16"), (_lib.BF16, "bf16")])
def test_rgb_tail_gather_kernel_shapes(cin, cout_tile, dt, name):
    """The RGB layer's own kernel (csrc/conv_rgb.hip: (tap, output) columns, nine-tap gather) takes 64..256 input
    channels in steps of 64, from weights packed for either cout tile; other widths (32, 80) stay on the direct
    kernel.  Several tiles in both directions, ragged right and bottom edges, two images."""
    n, h, w = 2, 37, 71
    x = q(rnd("x", (n, cin, h, w), 1), dt)
    wt = q(rnd("w", (3, cin, 3, 3), 2, 0.05), dt)
    b = rnd("b", (3,), 3) + 0.5
    core = F.conv2d(x, wt, b, padding=1)
    pc = ops.PackedConv(wt.numpy(), b.numpy(), dt, cout_tile=cout_tile)
    xp = ops.Planes.from_nchw(x, dt)
    mat = np.array([[1.0, 0.1, 0.0], [0.0, 0.9, 0.2], [0.3, 0.0, 1.1]], np.float32)
    tb = np.array([0.5, -0.25, 0.125], np.float32)
    ref = F.conv2d(core, torch.from_numpy(mat).view(3, 3, 1, 1), torch.from_numpy(tb))
    got = ops.conv3x3(xp, pc, tail=dict(mode=_lib.TAIL_F32_NCHW, mat=mat, bias=tb))
    assert_close(got.cpu(), ref, 5e-4, f"tail f32 {name} cin={cin}")
    got = ops.conv3x3(xp, pc, tail=dict(mode=_lib.TAIL_U8_NHWC, lo=0.0, hi=1.0, scale=255.0, round=_lib.ROUND_HALF_UP))
    ref8 = O.post_unit(core, O.ROUND_HALF_UP)
    diff = np.abs(got.cpu().numpy().astype(np.int32) - ref8.astype(np.int32))
    assert diff.max() <= 1, f"tail u8 {name} cin={cin}: max diff {diff.max()}"
    assert (diff != 0).mean() < 0.1


def test_u8_tail_matches_the_reference_python_postprocess():
    """The HIP u8 tail on exact .5 ties, near-ties and out-of-range values against the outputs of the reference's
    own postprocess / save_image functions (tests/golden/prepost_ref.npz).  An f32 convolution whose only non-zero
    weights are 1.0 on the centre tap of channels 0..2 hands the tensor to the tail unchanged."""
    from tests import helpers as H
    z = H.prepost_ref()
    t = torch.from_numpy(z["t"])
    x = torch.zeros((1, 8, t.shape[2], t.shape[3]))
    x[:, :3] = t
    wt = np.zeros((3, 8, 3, 3), np.float32)
    for c in range(3):
        wt[c, c, 1, 1] = 1.0
    pc = ops.PackedConv(wt, np.zeros(3, np.float32), _lib.F32, cout_tile=32)
    xp = ops.Planes.from_nchw(x, _lib.F32)
    for mode, key in [(_lib.ROUND_HALF_EVEN, "post_half_even_edsr_rknn"), (_lib.ROUND_HALF_EVEN, "post_half_even_rcan_rknn"),
                      (_lib.ROUND_TRUNC, "post_trunc_edsr_mtk")]:
        got = ops.conv3x3(xp, pc, tail=dict(mode=_lib.TAIL_U8_NHWC, lo=0.0, hi=255.0, scale=1.0, round=mode))
        assert np.array_equal(got.cpu().numpy()[0], z[key]), key
    got = ops.conv3x3(xp, pc, tail=dict(mode=_lib.TAIL_U8_NHWC, lo=0.0, hi=255.0, scale=1.0, round=_lib.ROUND_HALF_UP))
    assert np.array_equal(got.cpu().numpy(), O.post_range(t, O.ROUND_HALF_UP))


def test_pre_post_resize_match_the_reference_cpp_functions():
    """The fused u8 pre-processing of the first layer, the u8 tail in the runners' rounding mode (half-up) and the
    resize kernel against outputs of the reference C++ runners' own preprocess_image / postprocess_image /
    resize_image (tests/golden/prepost_cpp_ref.npz, tools/make_golden_prepost_cpp.py), bit for bit.  Convolutions
    with a single 1.0 on the centre tap hand the values through unchanged."""
    from tests import helpers as H
    z = H.prepost_cpp_ref()
    mean255 = (z["rgb_mean"] * np.float32(255.0)).astype(np.float32)
    # post-processing: f32 tensor -> identity conv -> u8 tail
    wt = np.zeros((3, 8, 3, 3), np.float32)
    for c in range(3):
        wt[c, c, 1, 1] = 1.0
    pc = ops.PackedConv(wt, np.zeros(3, np.float32), _lib.F32, cout_tile=32)

    def tail(t, **kw):
        x = torch.zeros((1, 8, t.shape[1], t.shape[2]))
        x[0, :3] = torch.from_numpy(t)
        return ops.conv3x3(ops.Planes.from_nchw(x, _lib.F32), pc,
                           tail=dict(mode=_lib.TAIL_U8_NHWC, round=_lib.ROUND_HALF_UP, **kw)).cpu().numpy()[0]

    for n in ("edsr", "rcan"):
        assert np.array_equal(tail(z["t"], lo=0.0, hi=255.0, scale=1.0), z[f"post_{n}_zero"]), n
        assert np.array_equal(tail(z["t"], lo=0.0, hi=255.0, scale=1.0, bias=mean255), z[f"post_{n}_mean"]), n
    assert np.array_equal(tail(z["t_unit"], lo=0.0, hi=1.0, scale=255.0), z["post_realesrgan"])
    # pre-processing: u8 image -> first layer whose output channel c is the centre tap of input channel c
    w0 = np.zeros((32, 3, 3, 3), np.float32)
    for c in range(3):
        w0[c, c, 1, 1] = 1.0
    img = torch.from_numpy(z["img"])[None].cuda()
    got = ops.conv_first(img, w0, np.zeros(32, np.float32), _lib.F32, pre_bias=-mean255).to_nchw().cpu().numpy()[0, :3]
    assert np.array_equal(got, z["pre_edsr"]) and np.array_equal(got, z["pre_rcan"])
    got = ops.conv_first(img, w0, np.zeros(32, np.float32), _lib.F32,
                         pre_mat=np.eye(3, dtype=np.float32) * np.float32(1.0 / 255.0)).to_nchw().cpu().numpy()[0, :3]
    assert np.array_equal(got, z["pre_realesrgan"])
    # resize
    for i, (ih, iw, oh, ow) in enumerate(z["resize_cases"].tolist()):
        src = synth.random_image(ih, iw, 300 + i)
        got = ops.resize_bilinear_u8(torch.from_numpy(src).cuda(), ow, oh).cpu().numpy()
        assert np.array_equal(got, z[f"resize_{i}"]), (ih, iw, oh, ow)


@pytest.mark.parametrize("dt,name,tol", DTYPES)
@pytest.mark.parametrize("mb", [2, 3])
def test_chansum(dt, name, tol, mb):
    n, h, w = 2, 35, 50
    x = q(rnd("x", (n, 64, h, w), 1), dt)
    wt = q(rnd("w", (64, 64, 3, 3), 2, 0.1), dt)
    b = rnd("b", (64,), 3)
    ref = F.conv2d(x, wt, b, padding=1)
    out, sums = ops.conv3x3(ops.Planes.from_nchw(x, dt), ops.PackedConv(wt.numpy(), b.numpy(), dt),
                            chansum=True, rows_per_wave=mb)
    got = sums.sum(dim=1).cpu()
    assert_close(got, ref.sum(dim=(2, 3)), 1e-4 if dt != _lib.F32 else 2e-5, f"chansum {name}")
    assert_close(out.to_nchw().cpu(), ref, tol)


@pytest.mark.parametrize("dt,name,tol", DTYPES)
def test_conv_first(dt, name, tol):
    n, h, w = 2, 17, 29
    img = synth.random_image(h, w, 3, n=n)
    wt = rnd("w", (64, 3, 3, 3), 2, 0.3)
    b = rnd("b", (64,), 3)
    # u8 path with the runners' /255 (realesrgan.cpp:28-48)
    ref = F.conv2d(O.pre_unit(img), wt, b, padding=1)
    out, out2 = ops.conv_first(torch.from_numpy(img).cuda(), wt.numpy(), b.numpy(), dt,
                               pre_mat=np.eye(3, dtype=np.float32) * np.float32(1.0 / 255.0), out2_dtype=_lib.F32)
    assert_close(out2.to_nchw().cpu(), ref, 2e-6, f"first u8 {name} f32 copy")
    assert_close(out.to_nchw().cpu(), ref, tol, f"first u8 {name}")
    # f32 NCHW path with MeanShift: padding must stay zero in the shifted domain (edsr_model.py:115)
    xin = O.pre_raw(img)
    mb_ = -255.0 * np.asarray(synth.RGB_MEAN, np.float32)
    ref = F.conv2d(xin + torch.from_numpy(mb_).view(1, 3, 1, 1), wt, b, padding=1)
    out = ops.conv_first(xin.cuda(), wt.numpy(), b.numpy(), _lib.F32, pre_bias=mb_)
    assert_close(out.to_nchw().cpu(), ref, 2e-6, "first f32 meanshift")


@pytest.mark.parametrize("dt,name,tol", DTYPES)
def test_channel_attention(dt, name, tol, golden_dir):
    # CALayer + RCAB residual against the reference modules' own outputs (tests/golden/ops.npz)
    g = np.load(f"{golden_dir}/ops.npz")
    sd = synth.rcan_state_dict(n_resgroups=1, n_resblocks=1, seed=22)
    p = "body.0.body.0.body.3.conv_du"
    x = torch.from_numpy(synth.hash_uniform("rcab_in", (2, 64, 11, 15), 22)) * 50.0
    xq = q(x, dt)
    xp = ops.Planes.from_nchw(xq, dt)
    sums = ops.channel_sums(xp)
    assert_close(sums.cpu(), xq.sum(dim=(2, 3)), 1e-5, "channel_sums")
    scale = ops.ca_scale(sums.view(2, 1, 64), 11 * 15, sd[p + ".0.weight"], sd[p + ".0.bias"],
                         sd[p + ".2.weight"], sd[p + ".2.bias"])
    zero = ops.Planes(2, 64, 11, 15, _lib.F32, zero=True)
    out = ops.ca_apply(xp, zero, scale, 1.0, out_dtype=_lib.F32)
    assert_close(out.to_nchw().cpu(), torch.from_numpy(g["ca_out"]), 2e-5 if dt == _lib.F32 else tol * 2, f"CALayer {name}")


@pytest.mark.parametrize("dt,name,tol", DTYPES)
def test_permutes(dt, name, tol, golden_dir):
    g = np.load(f"{golden_dir}/ops.npz")
    x = torch.from_numpy(synth.hash_uniform("ps_in", (1, 36, 5, 7), 23))
    # pad channels so that the shuffled tensor has whole 16-byte chunks: use 8x the channels
    xx = q(torch.cat([x] * 8, dim=1), dt)  # 288 channels -> 72 (r=2) / 32 (r=3)
    for r in (2, 3):
        ref = F.pixel_shuffle(xx, r)
        got = ops.pixel_shuffle(ops.Planes.from_nchw(xx, dt), r).to_nchw().cpu()
        assert torch.equal(got, ref), f"pixel_shuffle r={r} {name}"
    assert np.array_equal(F.pixel_shuffle(x, 2).numpy(), g["ps2_out"])
    x64 = q(rnd("near", (2, 64, 9, 14), 5), dt)
    got = ops.nearest_up2(ops.Planes.from_nchw(x64, dt)).to_nchw().cpu()
    assert torch.equal(got, O.nearest_up2(x64))
    assert np.array_equal(O.nearest_up2(x).numpy(), g["near_out"])


@pytest.mark.parametrize("dt,name", [(_lib.F16, "f16"), (_lib.BF16, "bf16")])
def test_aux_kernels_wide_paths(dt, name):
    """The 16-byte / 64-byte-per-thread forms of the standalone operators: PixelShuffle(2) on half tensors whose shuffled
    channel count is a multiple of 16, nearest x2 and the channel sums with several loads in flight per thread (long
    pixel segments), at ragged sizes and batch 2, against torch bit for bit (the sums: to f32 summation error)."""
    for (n, c_in, h, w) in [(2, 64, 7, 13), (1, 128, 33, 31), (2, 256, 17, 65), (1, 64, 1, 1)]:
        x = q(rnd(f"psw{c_in}", (n, c_in, h, w), 7), dt)
        got = ops.pixel_shuffle(ops.Planes.from_nchw(x, dt), 2).to_nchw().cpu()
        assert torch.equal(got, F.pixel_shuffle(x, 2)), f"pixel_shuffle wide {name} {(n, c_in, h, w)}"
    for (n, c, h, w) in [(2, 64, 9, 14), (1, 32, 31, 67), (1, 16, 1, 3)]:
        x = q(rnd(f"nrw{c}", (n, c, h, w), 8), dt)
        got = ops.nearest_up2(ops.Planes.from_nchw(x, dt)).to_nchw().cpu()
        assert torch.equal(got, O.nearest_up2(x)), f"nearest_up2 {name} {(n, c, h, w)}"
    for (n, c, h, w) in [(1, 64, 130, 257), (2, 32, 61, 90), (1, 16, 3, 5)]:
        x = q(rnd(f"csw{c}", (n, c, h, w), 9), dt)
        sums = ops.channel_sums(ops.Planes.from_nchw(x, dt)).cpu()
        ref = x.double().sum(dim=(2, 3))
        assert float((sums.double() - ref).abs().max()) <= 1e-5 * float(x.abs().double().sum(dim=(2, 3)).max()), f"channel_sums {name}"


def test_resize_bilinear_u8():
    for (ih, iw, oh, ow) in [(37, 53, 339, 510), (400, 600, 256, 256), (256, 256, 256, 256), (5, 7, 3, 2)]:
        img = synth.random_image(ih, iw, 7)
        ref = O.resize_bilinear_u8(img, ow, oh)
        got = ops.resize_bilinear_u8(torch.from_numpy(img).cuda(), ow, oh).cpu().numpy()
        assert np.array_equal(got, ref), f"resize {iw}x{ih}->{ow}x{oh}: {np.abs(got.astype(int) - ref).max()}"


def test_bad_arguments_fail_loudly():
    pc = ops.PackedConv(np.zeros((64, 64, 3, 3), np.float32), None, _lib.F16)
    x = ops.Planes(1, 64, 8, 8, _lib.F32)  # wrong dtype for an f16 conv
    with pytest.raises(_lib.SrmiError):
        ops.conv3x3(x, pc)
    with pytest.raises(_lib.SrmiError):
        ops.PackedConv(np.zeros((64, 50, 3, 3), np.float32), None, _lib.F16)  # cin not a plane multiple


def test_f16_stores_saturate_instead_of_overflowing():
    """Values beyond the half range are stored as +-65504, not +-inf (in-range values are unaffected)."""
    n, h, w = 1, 12, 32
    x = torch.full((n, 16, h, w), 100.0)
    wt = torch.zeros(32, 16, 3, 3)
    wt[:, :, 1, 1] = 50.0                     # centre tap only: 16 * 100 * 50 = 80000 > 65504
    wt[1] = -wt[1]
    wt[2] *= 0.5                              # 40000: representable
    pc = ops.PackedConv(wt.numpy(), np.zeros(32, np.float32), _lib.F16)
    out = ops.conv3x3(ops.Planes.from_nchw(x, _lib.F16), pc).to_nchw().cpu()
    assert torch.isfinite(out).all()
    assert float(out[0, 0].min()) == 65504.0 and float(out[0, 1].max()) == -65504.0
    assert float(out[0, 2].min()) == 40000.0 == float(out[0, 2].max())


def test_conv_output_rounded_to_8_significant_bits():
    """SRMI_TENSOR_ROUND8 on `out`: every stored f16 value is bfloat16-representable and equals the f32 result rounded
    to bfloat16 (nearest even); the plain store of the same conv differs from it."""
    n, h, w = 1, 25, 40
    x = q(rnd("x", (n, 64, h, w), 1), _lib.F16)
    wt = q(rnd("w", (32, 64, 3, 3), 2, 0.1), _lib.F16)
    b = rnd("b", (32,), 3)
    pc = ops.PackedConv(wt.numpy(), b.numpy(), _lib.F16)
    xp = ops.Planes.from_nchw(x, _lib.F16)
    exact = ops.conv3x3(xp, pc, act=_lib.ACT_LRELU, out_dtype=_lib.F32).to_nchw().cpu()
    plain = ops.conv3x3(xp, pc, act=_lib.ACT_LRELU).to_nchw().cpu()
    got = ops.conv3x3(xp, pc, act=_lib.ACT_LRELU, out_round8=True).to_nchw().cpu()
    want = exact.to(torch.bfloat16).to(torch.float16).to(torch.float32)
    assert torch.equal(got, want)
    assert torch.equal(got, got.to(torch.bfloat16).to(torch.float32))
    assert not torch.equal(got, plain)


@pytest.mark.parametrize("dt,name", [(_lib.F16, "f16"), (_lib.BF16, "bf16")])
@pytest.mark.parametrize("mb", [2, 3])
@pytest.mark.parametrize("c,cout", [(64, 64), (96, 32), (16, 64)])
def test_gated_input_equals_ca_apply_then_conv(dt, name, mb, c, cout):
    """A convolution reading x + alpha * (t * gate[c]) itself (RCAB output consumed by the next conv,
    rcan_model.py:94-97) gives the bytes of ca_apply followed by the plain convolution, and stores the same sum."""
    n, h, w = 2, 29, 71   # ragged tiles in both directions
    x = q(rnd("x", (n, c, h, w), 1), dt)
    t = q(rnd("t", (n, c, h, w), 2, 3.0), dt)
    gate = (rnd("g", (n, c), 3) * 0.5 + 0.5).cuda().contiguous()
    wt = q(rnd("w", (cout, c, 3, 3), 4, 0.1), dt)
    b = rnd("b", (cout,), 5)
    pc = ops.PackedConv(wt.numpy(), b.numpy(), dt)
    xp, tp = ops.Planes.from_nchw(x, dt), ops.Planes.from_nchw(t, dt)
    summed = ops.ca_apply(tp, xp, gate, 0.7)
    ref = ops.conv3x3(summed, pc, act=_lib.ACT_RELU, rows_per_wave=mb).to_nchw().cpu()
    in_sum = ops.Planes(n, c, h, w, dt, zero=True)
    got = ops.conv3x3(xp, pc, act=_lib.ACT_RELU, rows_per_wave=mb, in_gate=tp, gate=gate, gate_alpha=0.7, in_sum=in_sum)
    torch.cuda.synchronize()
    assert torch.equal(in_sum.to_nchw().cpu(), summed.to_nchw().cpu()), f"{name}: stored sum differs"
    assert torch.equal(got.to_nchw().cpu(), ref), f"{name}: gated conv differs from ca_apply + conv"
    # without the side store, and against the fp32 functional reference
    got2 = ops.conv3x3(xp, pc, act=_lib.ACT_RELU, rows_per_wave=mb, in_gate=tp, gate=gate, gate_alpha=0.7)
    assert torch.equal(got2.to_nchw().cpu(), ref)
    xs = q(x + 0.7 * (t * gate.cpu()[:, :, None, None]), dt)
    assert_close(ref, F.relu(F.conv2d(xs, wt, b, padding=1)), 2.0 ** -10 if dt == _lib.F16 else 2.0 ** -7, f"gated {name}")
    with pytest.raises(_lib.SrmiError):
        ops.conv3x3(xp, pc, in_gate=tp, gate=gate, in_sum=xp)   # the sum may not overwrite an input
    if mb == 2:
        # single-precision trunk: x and the stored sum are f32 tensors, the conv input is their half-precision cast
        x32 = rnd("x32", (n, c, h, w), 6)
        xp32 = ops.Planes.from_nchw(x32, _lib.F32)
        s16, s32 = ops.ca_apply(tp, xp32, gate, 0.7, out_dtype=dt, out2_dtype=_lib.F32)
        ref = ops.conv3x3(s16, pc, act=_lib.ACT_RELU, rows_per_wave=mb).to_nchw().cpu()
        sum32 = ops.Planes(n, c, h, w, _lib.F32, zero=True)
        got = ops.conv3x3(xp32, pc, act=_lib.ACT_RELU, rows_per_wave=mb, in_gate=tp, gate=gate, gate_alpha=0.7, in_sum=sum32)
        torch.cuda.synchronize()
        assert torch.equal(sum32.to_nchw().cpu(), s32.to_nchw().cpu()), f"{name}: stored f32 sum differs"
        assert torch.equal(got.to_nchw().cpu(), ref), f"{name}: gated conv with an f32 trunk differs"


@pytest.mark.parametrize("dt,name,tol", DTYPES)
@pytest.mark.parametrize("c,mb", [(64, 3), (64, 2), (96, 3), (128, 3)])
def test_attention_finished_inside_the_conv(dt, name, tol, c, mb):
    """Pooling + squeeze / excite + sigmoid done by the conv's last workgroup per image (CALayer,
    rcan_model.py:61-76) against the two-kernel form (tile sums, then srmi_ca_scale) and the functional reference;
    the accumulators are left at zero (ops.conv3x3 checks it), so a second launch gives the same bits."""
    n, h, w, cr = 3, 45, 70, max(c // 16, 1)
    x = q(rnd("x", (n, c, h, w), 1), dt)
    wt = q(rnd("w", (c, c, 3, 3), 2, 0.1), dt)
    b = rnd("b", (c,), 3)
    w1, b1 = rnd("w1", (cr, c), 4, 0.05), rnd("b1", (cr,), 5, 0.1)
    w2, b2 = rnd("w2", (c, cr), 6, 0.5), rnd("b2", (c,), 7, 0.1)
    pc = ops.PackedConv(wt.numpy(), b.numpy(), dt, cout_tile=32 if c % 64 else 64)
    xp = ops.Planes.from_nchw(x, dt)
    out, sums = ops.conv3x3(xp, pc, chansum=True, rows_per_wave=mb)
    two = ops.ca_scale(sums, h * w, w1.numpy(), b1.numpy(), w2.numpy(), b2.numpy()).cpu()
    out1, one = ops.conv3x3(xp, pc, rows_per_wave=mb, attention=(w1.numpy(), b1.numpy(), w2.numpy(), b2.numpy()))
    assert torch.equal(out1.to_nchw().cpu(), out.to_nchw().cpu())
    assert float((one.cpu() - two).abs().max()) <= 2e-6, "fixed-point pooling differs from the float tile sums"
    y = F.conv2d(x, wt, b, padding=1)
    ref = torch.sigmoid(F.linear(F.relu(F.linear(y.mean(dim=(2, 3)), w1, b1)), w2, b2))
    assert float((one.cpu() - ref).abs().max()) <= (1e-5 if dt == _lib.F32 else 2e-4)
    _, again = ops.conv3x3(xp, pc, rows_per_wave=mb, attention=(w1.numpy(), b1.numpy(), w2.numpy(), b2.numpy()))
    assert torch.equal(again.cpu(), one.cpu())


@pytest.mark.parametrize("dt,name,tol", DTYPES)
def test_conv_writes_only_its_channel_slice(dt, name, tol):
    """Out-of-bounds canary for the dense-block concat: a 32-output conv stores into channels [32, 64) of a wider
    plane buffer filled with a pattern; at sizes that are not multiples of the tile every other plane, and a guard
    band behind the buffer's last image, keep the pattern bit for bit."""
    for n, h, w in ((1, 13, 33), (2, 25, 65)):
        x = q(rnd("x", (n, 64, h, w), 1), dt)
        wt = q(rnd("w", (32, 64, 3, 3), 2, 0.1), dt)
        b = rnd("b", (32,), 3)
        pc = ops.PackedConv(wt.numpy(), b.numpy(), dt)
        out = ops.Planes(n + 1, 96, h, w, dt)          # one spare image behind the last one as the guard band
        out.buf.fill_(3.0)
        out.n = n
        ops.conv3x3(ops.Planes.from_nchw(x, dt), pc, act=_lib.ACT_LRELU, out=out, out_c0=32)
        torch.cuda.synchronize()
        cpb = ops.cpb(dt)
        lo, hi = 32 // cpb, 64 // cpb
        buf = out.buf.float()
        assert bool((buf[:n, :lo] == 3.0).all()) and bool((buf[:n, hi:] == 3.0).all()) and bool((buf[n] == 3.0).all()), \
            (name, n, h, w)
        out.n = n + 1
        got = out.to_nchw()[:n, 32:64].cpu()
        assert_close(got, F.leaky_relu(F.conv2d(x, wt, b, padding=1), 0.2), tol, f"slice conv {name}")
