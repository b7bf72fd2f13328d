"""Whole-network parity on the GPU, through the Python module surface -> C ABI -> HIP kernels.

 * f32 engine vs tests/golden (outputs of the reference's own modules): max-abs <= 1e-4 of the
   image scale.
 * f16 / bf16 engines vs the oracle on the u8 runner path: max u8 diff <= 1 LSB and PSNR >= 50 dB.
"""
import numpy as np
import pytest
import torch

from oracle import sr_oracle as O
from superresolution_amd import _lib, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", H.golden_cases())
def test_f32_matches_reference_golden(tag):
    meta, y_ref = H.load_case(tag)
    model, _ = H.build_model(meta, "f32")
    x = H.input_for(meta)
    y = model(x)                      # CPU tensor in -> CPU tensor out
    assert y.device.type == "cpu" and tuple(y.shape) == y_ref.shape
    err = O.max_abs(y.numpy(), y_ref)
    print(f"{tag}: max-abs {err:.3e} (tol {H.f32_tol(meta):.1e}, range {y_ref.min():.1f}..{y_ref.max():.1f})")
    assert err <= H.f32_tol(meta)


def _u8_case(arch):
    if arch == "rrdbnet":
        return dict(arch="rrdbnet", num_block=23, scale=4, seed=31, input="unit")
    if arch == "edsr":
        return dict(arch="edsr", n_resblocks=32, n_feats=256, scale=4, seed=32, input="range")
    return dict(arch="rcan", n_resgroups=10, n_resblocks=20, scale=4, seed=33, input="range")


@pytest.mark.parametrize("arch", ["rrdbnet", "edsr", "rcan"])
@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
def test_u8_runner_path_full_depth(arch, dtype):
    """Full-depth networks on a smooth 48x64 image (u8 in -> u8 out, runner pre/post fused)."""
    meta = _u8_case(arch)
    h, w = 48, 64
    img = synth.smooth_image(h, w, meta["seed"])
    model, sd = H.build_model(meta, dtype)
    got = model.forward_u8(img, _lib.ROUND_HALF_UP)
    assert got.dtype == torch.uint8 and tuple(got.shape) == (h * 4, w * 4, 3)
    with torch.no_grad():
        if arch == "rrdbnet":
            ref = O.post_unit(H.oracle_forward(meta, sd, O.pre_unit(img)), O.ROUND_HALF_UP)[0]
        else:
            # runner semantics: p - mean*range ... + mean*range == the full model on raw 0..255 input
            ref = O.post_range(H.oracle_forward(meta, sd, O.pre_raw(img)), O.ROUND_HALF_UP)[0]
    g = got.numpy()
    diff = np.abs(g.astype(np.int32) - ref.astype(np.int32))
    psnr = O.psnr_u8(g, ref)
    print(f"{arch} {dtype}: max u8 diff {diff.max()}, differing {100 * (diff != 0).mean():.3f}%, PSNR {psnr:.2f} dB, "
          f"clamped lo {100 * (ref == 0).mean():.1f}% hi {100 * (ref == 255).mean():.1f}%")
    # One bound for every dtype: u8 diff <= 1 LSB and PSNR >= 50 dB.  f32 / f16 meet it.  bf16 keeps only 8
    # significant bits per conv operand: through 351 (RRDBNet) / 411 (RCAN) convolutions it stays above 50 dB but
    # a few pixels move by 2-3 LSB, so f16 is the default half type and a bf16 run that misses the bound is
    # reported as an expected failure with its measured error, never as a pass against a looser bound.
    if dtype == "bf16" and (diff.max() > 1 or psnr < 50.0):
        pytest.xfail(f"bf16 operands miss the half-precision bound on {arch}: max u8 diff {diff.max()} LSB "
                     f"(> 1 LSB on {100 * (diff > 1).mean():.4f} % of the samples), PSNR {psnr:.2f} dB")
    assert diff.max() <= 1
    assert psnr >= 50.0


def test_f16_without_f32_trunk_is_reported():
    """trunk_f32=False stores the residual trunk in f16 too (less HBM traffic); accuracy is
    reported and only loosely bounded."""
    meta = _u8_case("rrdbnet")
    img = synth.smooth_image(48, 64, meta["seed"])
    model, sd = H.build_model(meta, "f16", trunk_f32=False)
    got = model.forward_u8(img, _lib.ROUND_HALF_UP).numpy()
    with torch.no_grad():
        ref = O.post_unit(H.oracle_forward(meta, sd, O.pre_unit(img)), O.ROUND_HALF_UP)[0]
    diff = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    print(f"rrdbnet f16 trunk_f32=False: max u8 diff {diff.max()}, PSNR {O.psnr_u8(got, ref):.2f} dB")
    assert diff.max() <= 1 and O.psnr_u8(got, ref) >= 50.0


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("scale", [4, 2])
def test_rrdbnet_upsampling_convs_in_both_forms(dtype, scale):
    """conv_up1 / conv_up2 (behind the nearest-x2 upsamplings, realesrgan_model.py:115-116) run as four 2x2 convolutions
    on the source grid by default and as nine-tap convolutions with set_subpixel_up(False): the two results lie next to
    each other (the summed taps are rounded once more) and within the half-precision bound of the oracle.  The engine
    uses the sub-pixel form with f16 compute only: with bf16 the switch changes nothing."""
    meta = dict(arch="rrdbnet", num_block=2, scale=scale, seed=31, input="unit")
    model, sd = H.build_model(meta, dtype, trunk_f32=False)
    eng = model.engine()
    assert eng.subpixel_up()
    imgs = np.stack([synth.smooth_image(37, 53, 5), synth.random_image(37, 53, 6)])
    with torch.no_grad():
        ref = O.post_unit(H.oracle_forward(meta, sd, O.pre_unit(imgs)), O.ROUND_HALF_UP)
    outs = {}
    for on in (True, False):
        eng.set_subpixel_up(on)
        outs[on] = model.forward_u8(imgs, _lib.ROUND_HALF_UP).numpy()
        d = np.abs(outs[on].astype(np.int32) - ref.astype(np.int32))
        print(f"rrdbnet x{scale} {dtype} sub-pixel upsampling convs={on}: max u8 diff {d.max()}, PSNR {O.psnr_u8(outs[on], ref):.2f} dB")
        if dtype == "f16":
            assert d.max() <= 1 and O.psnr_u8(outs[on], ref) >= 50.0
    eng.set_subpixel_up(True)
    d12 = np.abs(outs[True].astype(np.int32) - outs[False].astype(np.int32))
    if dtype == "bf16":
        assert d12.max() == 0
    else:
        assert d12.max() <= 1 and (d12 != 0).mean() < 0.05 and (d12 != 0).any(), (d12.max(), (d12 != 0).mean())


@pytest.mark.parametrize("trunk_f32", [True, False])
def test_bench_configuration_full_size_frame(trunk_f32):
    """The benchmark's exact workload (RealESRGAN_x4plus, 23 blocks, 510x339 -> 2040x1356, f16)
    against the oracle on one smooth and one uniform-noise frame."""
    meta = dict(arch="rrdbnet", num_block=23, scale=4, seed=0, input="unit")
    model, sd = H.build_model(meta, "f16", trunk_f32=trunk_f32)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    for kind, img in (("smooth", synth.smooth_image(339, 510, 0)), ("noise", synth.random_image(339, 510, 100))):
        got = model.forward_u8(img, _lib.ROUND_HALF_UP).numpy()
        with torch.no_grad():
            ref = O.post_unit(H.oracle_forward(meta, sd, O.pre_unit(img)), O.ROUND_HALF_UP)[0]
        diff = np.abs(got.astype(np.int32) - ref.astype(np.int32))
        print(f"510x339 {kind} trunk_f32={trunk_f32}: max u8 diff {diff.max()}, differing {100 * (diff != 0).mean():.2f}%, "
              f"PSNR {O.psnr_u8(got, ref):.2f} dB, clamped lo {100 * (ref == 0).mean():.1f}% hi {100 * (ref == 255).mean():.1f}%")
        assert got.shape == (1356, 2040, 3)
        assert diff.max() <= 1 and O.psnr_u8(got, ref) >= 50.0


def test_edsr_batch_configuration_full_size():
    """EDSR x4 (32 blocks x 256 features) on 256x256 frames: frame 0 of a batch against the oracle, and the
    frames of a 32-frame batch against the same frames run alone (bit for bit)."""
    meta = _u8_case("edsr")
    model, sd = H.build_model(meta, "f16")
    torch.set_num_threads(min(16, torch.get_num_threads()))
    imgs = np.stack([synth.smooth_image(256, 256, 40)] + [synth.random_image(256, 256, 41 + i) for i in range(3)])
    got = model.forward_u8(imgs, _lib.ROUND_HALF_UP).numpy()
    assert got.shape == (4, 1024, 1024, 3)
    with torch.no_grad():
        ref = O.post_range(H.oracle_forward(meta, sd, O.pre_raw(imgs[0])), O.ROUND_HALF_UP)[0]
    diff = np.abs(got[0].astype(np.int32) - ref.astype(np.int32))
    print(f"edsr 256x256 f16: max u8 diff {diff.max()}, PSNR {O.psnr_u8(got[0], ref):.2f} dB")
    assert diff.max() <= 1 and O.psnr_u8(got[0], ref) >= 50.0
    eng = model.engine()
    batch = torch.from_numpy(np.concatenate([imgs] * 8)).cuda()          # 32 frames
    out = eng.forward_u8(batch).cpu().numpy()
    for i in range(32):
        assert np.array_equal(out[i], got[i % 4]), i


def test_edsr_batch_configuration_full_size_bf16():
    """The configuration the baseline names in bf16 (EDSR x4, 256x256, batch 32), run in bf16 at full size and
    depth against the half-precision bound (<= 1 LSB, >= 50 dB).  With 8 significant bits per conv operand in the
    0..255 domain a few pixels move by 2 LSB while the PSNR stays above 50 dB: the run is then an EXPECTED FAILURE
    whose reason carries the measured error.  The bound is met by the f16 engine (test above)."""
    meta = _u8_case("edsr")
    model, sd = H.build_model(meta, "bf16")
    torch.set_num_threads(min(16, torch.get_num_threads()))
    imgs = np.stack([synth.smooth_image(256, 256, 40), synth.random_image(256, 256, 41)])
    batch = torch.from_numpy(np.concatenate([imgs] * 16)).cuda()          # 32 frames
    out = model.engine().forward_u8(batch).cpu().numpy()
    assert out.shape == (32, 1024, 1024, 3)
    worst = []
    for i, kind in enumerate(("smooth", "noise")):
        with torch.no_grad():
            ref = O.post_range(H.oracle_forward(meta, sd, O.pre_raw(imgs[i])), O.ROUND_HALF_UP)[0]
        diff = np.abs(out[i].astype(np.int32) - ref.astype(np.int32))
        psnr = O.psnr_u8(out[i], ref)
        print(f"edsr 256x256 batch 32 bf16 ({kind}): max u8 diff {diff.max()}, > 1 LSB on {100 * (diff > 1).mean():.4f} % of "
              f"the samples, differing {100 * (diff != 0).mean():.2f} %, PSNR {psnr:.2f} dB")
        worst.append((kind, int(diff.max()), float((diff > 1).mean()), float(psnr)))
    for i in range(32):
        assert np.array_equal(out[i], out[i % 2]), i
    if any(m > 1 or ps < 50.0 for _, m, _, ps in worst):
        pytest.xfail("EDSR x4 256x256 batch 32 in bf16 misses the <= 1 LSB / >= 50 dB bound: " +
                     "; ".join(f"{k}: max {m} LSB, > 1 LSB on {100 * f:.4f} %, {ps:.2f} dB" for k, m, f, ps in worst))


def test_rcan_configuration_full_size_frame():
    """RCAN x4 (10 groups x 20 blocks) on a 510x339 frame against the oracle: the channel-attention means run
    over 172 890 pixels in 464 tile partial sums."""
    meta = _u8_case("rcan")
    model, sd = H.build_model(meta, "f16")
    torch.set_num_threads(min(16, torch.get_num_threads()))
    img = synth.smooth_image(339, 510, 50)
    got = model.forward_u8(img, _lib.ROUND_HALF_UP).numpy()
    with torch.no_grad():
        ref = O.post_range(H.oracle_forward(meta, sd, O.pre_raw(img)), O.ROUND_HALF_UP)[0]
    diff = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    print(f"rcan 510x339 f16: max u8 diff {diff.max()}, PSNR {O.psnr_u8(got, ref):.2f} dB")
    assert got.shape == (1356, 2040, 3)
    assert diff.max() <= 1 and O.psnr_u8(got, ref) >= 50.0
    # The attention output is formed by the consuming convolution (gated input), from an f16 trunk (the benchmark's
    # RCAN configuration) or an f32 one (the default); the separate elementwise pass must give the same bytes.
    frames = torch.from_numpy(np.stack([img, synth.random_image(339, 510, 51)])).cuda()
    for trunk_f32 in (False, True):
        m2, _ = H.build_model(meta, "f16", trunk_f32=trunk_f32)
        eng = m2.engine()
        # The default is the one-kernel block (both convolutions of a block in one launch), with either trunk: checked
        # against the oracle here; its pooled sums see other tile shapes than the two-launch form, so the byte identity
        # below is asserted between the two forms of the two-launch path.
        one_kernel = eng.forward_u8(frames, _lib.ROUND_HALF_UP).cpu().numpy()
        d1 = np.abs(one_kernel[0].astype(np.int32) - ref.astype(np.int32))
        print(f"rcan 510x339 f16 trunk_f32={trunk_f32}, one-kernel blocks: max u8 diff {d1.max()}, PSNR {O.psnr_u8(one_kernel[0], ref):.2f} dB")
        assert eng.fused_rcab() and d1.max() <= 1 and O.psnr_u8(one_kernel[0], ref) >= 50.0
        if trunk_f32:
            assert np.array_equal(one_kernel[0], got)      # the module's own forward ran this plan
        eng.set_fused_rcab(False)
        fused = eng.forward_u8(frames, _lib.ROUND_HALF_UP).cpu().numpy()
        eng.set_fused_attention(False)
        plain = eng.forward_u8(frames, _lib.ROUND_HALF_UP).cpu().numpy()
        assert np.array_equal(fused, plain), f"trunk_f32={trunk_f32}"
        diff = np.abs(fused[0].astype(np.int32) - ref.astype(np.int32))
        print(f"rcan 510x339 f16 trunk_f32={trunk_f32}: max u8 diff {diff.max()}, PSNR {O.psnr_u8(fused[0], ref):.2f} dB")
        assert diff.max() <= 1 and O.psnr_u8(fused[0], ref) >= 50.0


def test_1080p_frame_whole_and_halo_tiled():
    """Real-ESRGAN x4plus on a 1920x1080 frame -> 7680x4320: the whole frame in one call, and the same frame as
    a 2x4 grid of halo-overlapped tiles (the multi-GPU tile mode).  The oracle is too slow at this size, so the
    test uses size-independent properties: a repeated run is bit-identical, the tiled result stays within a
    bounded error of the whole frame (the 32-px halo is far below the 348-px receptive field, so seams are
    approximate by design), and a corner block far from every seam agrees exactly."""
    from superresolution_amd import sharding
    meta = dict(arch="rrdbnet", num_block=23, scale=4, seed=0, input="unit")
    model, _ = H.build_model(meta, "f16", trunk_f32=False)
    eng = model.engine()
    img = torch.from_numpy(synth.smooth_image(1080, 1920, 7)).cuda()
    whole = eng.forward_u8(img)
    assert tuple(whole.shape) == (4320, 7680, 3)
    again = eng.forward_u8(img)
    assert torch.equal(whole, again)
    tiled = sharding.upscale_tiled(lambda t: eng.forward_u8(t), img, 4, 540, 480, 32)
    assert tuple(tiled.shape) == (4320, 7680, 3)
    # tiles of equal crop shape sent through the lanes together: the same frame, bit for bit
    assert torch.equal(sharding.upscale_tiled_batched(lambda t: eng.forward_u8(t), img, 4, 540, 480, 32), tiled)
    w_np, t_np = whole.cpu().numpy(), tiled.cpu().numpy()
    d = np.abs(w_np.astype(np.int32) - t_np.astype(np.int32))
    psnr = O.psnr_u8(w_np, t_np)
    print(f"1080p halo 32: max diff {d.max()}, differing {100 * (d != 0).mean():.3f}%, PSNR {psnr:.1f} dB")
    assert psnr >= 45.0
    # the frame's outer border band belongs to border tiles whose own borders coincide with the frame's: padding
    # is identical there, so the top-left corner block (far from any seam) must agree exactly
    assert np.array_equal(w_np[:256, :256], t_np[:256, :256])


@pytest.mark.parametrize("arch", ["rrdbnet", "edsr", "rcan"])
def test_edge_inputs_constant_frames_and_tiny_sizes(arch):
    """Constant black / white frames (every pixel clamps or sits on the border of the u8 range) and
    degenerate sizes (1x1, one row, one column, one pixel more than a tile) against the oracle, f16."""
    meta = _u8_case(arch)
    if arch == "rrdbnet":
        meta = dict(meta, num_block=2)
    elif arch == "edsr":
        meta = dict(meta, n_resblocks=3, n_feats=64)
    else:
        meta = dict(meta, n_resgroups=2, n_resblocks=2)
    model, sd = H.build_model(meta, "f16")
    cases = [np.zeros((24, 40, 3), np.uint8), np.full((24, 40, 3), 255, np.uint8)]
    cases += [synth.random_image(h, w, 60 + i) for i, (h, w) in enumerate([(1, 1), (1, 37), (29, 1), (13, 33), (2, 2)])]
    for img in cases:
        got = model.forward_u8(img, _lib.ROUND_HALF_UP).numpy()
        with torch.no_grad():
            if arch == "rrdbnet":
                ref = O.post_unit(H.oracle_forward(meta, sd, O.pre_unit(img)), O.ROUND_HALF_UP)[0]
            else:
                ref = O.post_range(H.oracle_forward(meta, sd, O.pre_raw(img)), O.ROUND_HALF_UP)[0]
        assert got.shape == ref.shape == (img.shape[0] * 4, img.shape[1] * 4, 3)
        diff = np.abs(got.astype(np.int32) - ref.astype(np.int32))
        assert diff.max() <= 1, (arch, img.shape, int(diff.max()))


@pytest.mark.parametrize("shape", ["heavy_tailed", "sparse"])
def test_half_precision_bound_on_other_weight_distributions(shape):
    """The f16 margins elsewhere in this file are measured on uniformly distributed weights.  No real checkpoint is
    available, so this is the next best thing: the same full-depth RRDBNet with the dense-block weights reshaped to a
    heavy-tailed distribution (sign(w) |w|^3, a few weights carry most of the energy, as trained convolutions do) and to
    a sparse one (four of five weights zero), both rescaled to the original RMS per tensor.  The bound is the same:
    u8 diff <= 1 LSB and PSNR >= 50 dB against the fp32 oracle, with the two-precision dense blocks on and off."""
    from superresolution_amd import RRDBNet
    meta = dict(arch="rrdbnet", num_block=23, scale=4, seed=61, input="unit")
    sd = H.state_dict_for(meta)
    for k in list(sd):
        if ".rdb" in k and k.endswith(".weight"):
            w = sd[k].astype(np.float64)
            rms = np.sqrt(np.mean(w * w))
            if shape == "heavy_tailed":
                v = np.sign(w) * np.abs(w / np.abs(w).max()) ** 3
            else:
                keep = synth.hash_uniform(k + ".mask", w.shape, 5) > 0.3      # uniform in [-0.5, 0.5): one in five kept
                v = np.where(keep, w, 0.0)
            sd[k] = (v * (rms / max(np.sqrt(np.mean(v * v)), 1e-30))).astype(np.float32)
    img = synth.smooth_image(40, 56, 13)
    with torch.no_grad():
        ref = O.post_unit(H.oracle_forward(meta, sd, O.pre_unit(img)), O.ROUND_HALF_UP)[0]
    assert ref.min() < 8 and ref.max() > 247 and 40 < ref.mean() < 215      # the frame still uses the whole range
    for round8 in (True, False):
        model = RRDBNet(num_block=23, scale=4, compute_dtype="f16", growth_round8=round8)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        got = model.forward_u8(img, _lib.ROUND_HALF_UP).numpy()
        d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
        psnr = O.psnr_u8(got, ref)
        print(f"{shape} weights, two-precision dense blocks {round8}: max u8 diff {d.max()}, PSNR {psnr:.1f} dB")
        assert d.max() <= 1 and psnr >= 50.0, (shape, round8, int(d.max()), psnr)


@pytest.mark.parametrize("arch", ["rrdbnet", "edsr", "rcan"])
def test_outputs_stay_inside_their_buffers(arch):
    """Out-of-bounds canaries: the u8 and f32 results are written into the middle of a larger buffer filled with a
    pattern; after forwards at sizes that are not multiples of the 12 x 32 tile (and of the x2 / x3 / x4 shuffles)
    every byte outside the result still holds the pattern, and the result equals the one written to a buffer of its own."""
    meta = _u8_case(arch)
    if arch == "rrdbnet":
        meta = dict(meta, num_block=1)
    elif arch == "edsr":
        meta = dict(meta, n_resblocks=2, n_feats=64, scale=3)
    else:
        meta = dict(meta, n_resgroups=1, n_resblocks=2, scale=2)
    model, _ = H.build_model(meta, "f16")
    eng = model.engine()
    s = eng.scale
    guard = 1 << 16
    for n, h, w in ((1, 13, 33), (2, 37, 51), (1, 1, 65)):
        img = torch.from_numpy(synth.random_image(h, w, 90 + h, n=n)).cuda()
        want = eng.forward_u8(img)
        size = n * h * s * w * s * 3
        big = torch.full((size + 2 * guard,), 0xA5, dtype=torch.uint8, device="cuda")
        got = eng.forward_u8(img, out=big[guard:guard + size].view(n, h * s, w * s, 3))
        torch.cuda.synchronize()
        assert torch.equal(got, want)
        assert bool((big[:guard] == 0xA5).all()) and bool((big[guard + size:] == 0xA5).all()), (arch, n, h, w)
        x = torch.from_numpy(synth.random_image(h, w, 91 + h, n=n)).permute(0, 3, 1, 2).float().cuda()
        if arch == "rrdbnet":
            x = x / 255.0
        wantf = eng.forward_f32(x)
        bigf = torch.full((size + 2 * guard,), -12345.0, dtype=torch.float32, device="cuda")
        gotf = eng.forward_f32(x, out=bigf[guard:guard + size].view(n, 3, h * s, w * s))
        torch.cuda.synchronize()
        assert torch.equal(gotf, wantf)
        assert bool((bigf[:guard] == -12345.0).all()) and bool((bigf[guard + size:] == -12345.0).all()), (arch, n, h, w)


def test_one_model_driven_from_two_threads_on_two_streams():
    """Calls on one model share its workspace: the library orders them (a mutex around the enqueue, an event between
    consecutive calls whatever stream they run on), so two threads with a stream each get the results a single
    thread gets."""
    import threading
    meta = dict(arch="rcan", n_resgroups=1, n_resblocks=2, scale=4, seed=8, input="range")   # RCAN: shared sums / scale too
    model, _ = H.build_model(meta, "f16")
    eng = model.engine()
    imgs = [torch.from_numpy(synth.random_image(37, 52, 80 + i, n=2)).cuda() for i in range(2)]
    want = [eng.forward_u8(x).clone() for x in imgs]
    torch.cuda.synchronize()
    errors = []

    def worker(i):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(8):
                    got = eng.forward_u8(imgs[i])
                    st.synchronize()
                    if not torch.equal(got, want[i]):
                        errors.append(f"thread {i}: result differs")
                        return
        except Exception as e:  # noqa: BLE001
            errors.append(f"thread {i}: {e!r}")

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_empty_batch_gives_an_empty_result():
    """torch's conv2d / interpolate accept a batch of zero images, so the reference modules return an empty tensor
    of the output shape; so do the f32, u8 and host-buffer paths here (nothing is launched)."""
    meta = dict(arch="rrdbnet", num_block=1, scale=4, seed=3, input="unit")
    model, sd = H.build_model(meta, "f16")
    with torch.no_grad():
        ref = H.oracle_forward(meta, sd, torch.zeros(0, 3, 9, 11))
    assert tuple(ref.shape) == (0, 3, 36, 44)
    y = model(torch.zeros(0, 3, 9, 11))
    assert tuple(y.shape) == (0, 3, 36, 44) and y.dtype == torch.float32
    eng = model.engine()
    assert tuple(eng.forward_u8(torch.zeros(0, 9, 11, 3, dtype=torch.uint8).cuda()).shape) == (0, 36, 44, 3)
    assert eng.forward_u8_host(np.zeros((0, 9, 11, 3), np.uint8)).shape == (0, 36, 44, 3)
    # zero-sized frames are an error in the reference (conv2d rejects them) and here
    with pytest.raises(Exception):
        model(torch.zeros(1, 3, 0, 11))


def test_halo_tiling_is_exact_when_halo_covers_the_receptive_field():
    """Tiles (+halo) of a frame upscaled independently and stitched == the whole frame upscaled,
    bit for bit, once the halo covers the receptive field (1 RRDB: 15 convs + conv_body + tail)."""
    from superresolution_amd import sharding
    meta = dict(arch="rrdbnet", num_block=1, scale=4, seed=11, input="unit")
    model, _ = H.build_model(meta, "f16")
    eng = model.engine()
    img = torch.from_numpy(synth.smooth_image(96, 130, 11)).cuda()
    whole = eng.forward_u8(img)
    tiled = sharding.upscale_tiled(lambda t: eng.forward_u8(t), img, 4, 48, 64, 24)
    assert torch.equal(whole, tiled)
    small_halo = sharding.upscale_tiled(lambda t: eng.forward_u8(t), img, 4, 48, 64, 4)
    d = (whole.int() - small_halo.int()).abs()
    print(f"halo 4: max seam diff {int(d.max())}, PSNR {O.psnr_u8(whole.cpu().numpy(), small_halo.cpu().numpy()):.1f} dB")


def test_batch_and_sweep_are_bit_identical():
    meta = dict(arch="rrdbnet", num_block=2, scale=4, seed=5, input="unit")
    model, _ = H.build_model(meta, "f16")
    imgs = torch.from_numpy(synth.random_image(20, 36, 9, n=3)).cuda()
    eng = model.engine()
    a = eng.forward_u8(imgs).cpu()
    eng.set_sweep(2)
    b = eng.forward_u8(imgs).cpu()
    eng.set_sweep(3)
    c = eng.forward_u8(imgs).cpu()
    one = torch.stack([eng.forward_u8(imgs[i]).cpu() for i in range(3)])
    assert torch.equal(a, b) and torch.equal(a, c) and torch.equal(a, one)
    # sweeps spread over 1..4 internal streams, repeated so that a lane's workspace is reused
    eng.set_sweep(1)
    imgs7 = torch.from_numpy(synth.random_image(20, 36, 10, n=7)).cuda()
    eng.set_lanes(1)
    ref = eng.forward_u8(imgs7).cpu()
    for lanes in (2, 3, 4):
        eng.set_lanes(lanes)
        for _ in range(2):
            assert torch.equal(eng.forward_u8(imgs7).cpu(), ref), lanes
    with pytest.raises(Exception):
        eng.set_lanes(5)


def test_core_plus_manual_meanshift_equals_full():
    """The reference's own identity (edsr/mtk/python/test/test_pytorch.py:137-198, tol 1e-3):
    full(x) == core(x - 255*mean) + 255*mean, here for the HIP EDSR and RCAN in f32."""
    from superresolution_amd import edsr_model, rcan_model
    mean = torch.tensor(synth.RGB_MEAN).view(1, 3, 1, 1) * 255.0
    x = torch.randn(1, 3, 24, 28, generator=torch.Generator().manual_seed(0)) * 127.5 + 127.5
    sd = {k: torch.from_numpy(v) for k, v in synth.edsr_state_dict(n_resblocks=3, n_feats=64, seed=3).items()}
    full = edsr_model.EDSR(n_resblocks=3, n_feats=64, compute_dtype="f32")
    full.load_state_dict(sd)
    core = edsr_model.create_core_model_from_full(full, scale=4)
    assert float((full(x) - (core(x - mean) + mean)).abs().max()) < 1e-3
    sd = {k: torch.from_numpy(v) for k, v in synth.rcan_state_dict(n_resgroups=2, n_resblocks=2, seed=4).items()}
    full = rcan_model.RCAN(scale=4, n_resgroups=2, n_resblocks=2, compute_dtype="f32")
    full.load_state_dict(sd)
    core = rcan_model.create_core_model_from_full(full, scale=4)
    assert float((full(x) - (core(x - mean) + mean)).abs().max()) < 1e-3


def test_weights_reupload_after_change():
    meta = dict(arch="rrdbnet", num_block=1, scale=2, seed=5, input="unit")
    model, sd = H.build_model(meta, "f32")
    x = torch.from_numpy(synth.hash_uniform("x", (1, 3, 12, 12), 1)) * 0.5 + 0.5
    y0 = model(x)
    with torch.no_grad():
        model.conv_last.bias.add_(0.25)
    y1 = model(x)
    assert float((y1 - y0 - 0.25).abs().max()) < 1e-5


def test_engine_rejects_bad_input():
    meta = dict(arch="rrdbnet", num_block=1, scale=4, seed=5, input="unit")
    model, _ = H.build_model(meta, "f16")
    with pytest.raises(ValueError):
        model(torch.zeros(1, 4, 8, 8))
    eng = model.engine()
    with pytest.raises(_lib.SrmiError):
        eng.load_state_dict({"not.a.key": np.zeros(3, np.float32)})


@pytest.mark.parametrize("n_feats,scale", [(96, 2), (32, 3), (160, 4)])
def test_edsr_widths_that_are_not_multiples_of_64(n_feats, scale):
    """Feature counts that are multiples of 32 but not of 64 (and scale 3 with 32 features: a 288-channel upsampler
    conv) run on 32-wide output tiles and match the oracle like every other shape."""
    meta = dict(arch="edsr", n_resblocks=2, n_feats=n_feats, scale=scale, seed=91, input="range", n=1, h=19, w=27)
    model, sd = H.build_model(meta, "f32")
    x = H.input_for(meta)
    y = model(x).numpy()
    with torch.no_grad():
        ref = H.oracle_forward(meta, sd, x).numpy()
    err = np.abs(y - ref).max()
    print(f"edsr n_feats={n_feats} x{scale}: max-abs {err:.3e}")
    assert y.shape == (1, 3, 19 * scale, 27 * scale) and err <= H.f32_tol(meta)


@pytest.mark.parametrize("n_feats,reduction", [(96, 16), (32, 8), (128, 16)])
def test_rcan_widths_and_fused_attention(n_feats, reduction):
    """RCAN at feature counts other than 64: 32-wide output tiles (three channel tiles per image take tickets in the
    attention tail), two-stage and eight-stage gated convolutions.  f32 against the oracle; f16 with either trunk
    precision gives the same bytes with and without the gated input, in one batch and frame by frame."""
    from superresolution_amd import RCAN
    n_groups, n_blocks, scale, h, w = 2, 3, 2, 27, 41
    sd = synth.rcan_state_dict(scale=scale, n_resgroups=n_groups, n_resblocks=n_blocks, n_feats=n_feats, reduction=reduction, seed=93)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    kw = dict(scale=scale, n_resgroups=n_groups, n_resblocks=n_blocks, n_feats=n_feats, reduction=reduction)
    imgs = synth.random_image(h, w, 94, n=3)
    x = O.pre_raw(imgs)
    m32 = RCAN(compute_dtype="f32", **kw)
    m32.load_state_dict(tsd, strict=True)
    with torch.no_grad():
        ref = O.rcan_forward(sd, x, scale=scale, n_resgroups=n_groups, n_resblocks=n_blocks, res_scale=1.0).numpy()
    err = np.abs(m32(x).numpy() - ref).max()
    print(f"rcan n_feats={n_feats}: f32 max-abs {err:.3e}")
    assert err <= 1e-3
    ref_u8 = O.post_range(torch.from_numpy(ref), O.ROUND_HALF_UP)
    for trunk_f32 in (False, True):
        m = RCAN(compute_dtype="f16", trunk_f32=trunk_f32, **kw)
        m.load_state_dict(tsd, strict=True)
        eng = m.engine()
        batch = torch.from_numpy(imgs).cuda()
        fused = eng.forward_u8(batch, _lib.ROUND_HALF_UP).cpu().numpy()
        one_by_one = np.concatenate([eng.forward_u8(batch[i:i + 1], _lib.ROUND_HALF_UP).cpu().numpy() for i in range(3)])
        eng.set_fused_attention(False)
        plain = eng.forward_u8(batch, _lib.ROUND_HALF_UP).cpu().numpy()
        assert np.array_equal(fused, plain) and np.array_equal(fused, one_by_one), f"trunk_f32={trunk_f32}"
        diff = np.abs(fused.astype(np.int32) - ref_u8.astype(np.int32))
        assert diff.max() <= 1 and O.psnr_u8(fused, ref_u8) >= 50.0


def test_caller_supplied_out_tensor_is_validated():
    """`out=` goes to the kernels as a raw pointer: a tensor of another shape, type or layout must be refused."""
    meta = dict(arch="rrdbnet", num_block=1, scale=4, seed=92, input="unit")
    model, _ = H.build_model(meta, "f16")
    eng = model.engine()
    img = torch.from_numpy(synth.random_image(16, 24, 1, n=1)).cuda()
    good = torch.empty((1, 64, 96, 3), dtype=torch.uint8, device="cuda")
    assert eng.forward_u8(img, out=good) is good
    for bad in (torch.empty((1, 64, 96, 4), dtype=torch.uint8, device="cuda")[..., :3],    # strided
                torch.empty((1, 64, 95, 3), dtype=torch.uint8, device="cuda"),             # wrong shape
                torch.empty((1, 64, 96, 3), dtype=torch.int8, device="cuda"),              # wrong type
                torch.empty((1, 64, 96, 3), dtype=torch.uint8)):                           # wrong device
        with pytest.raises(ValueError):
            eng.forward_u8(img, out=bad)
    with pytest.raises(ValueError):
        eng.forward_f32(torch.rand(1, 3, 16, 24), out=torch.empty((1, 3, 64, 96), dtype=torch.float16, device="cuda"))
