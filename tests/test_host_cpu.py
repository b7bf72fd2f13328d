"""CPU tests of the host side: module surface (constructor signatures, state_dict layout, loader
semantics, file-name heuristics), the C ABI's exported symbols, host-side weight packing, and the
fail-loudly behaviour without a GPU.  No kernel is launched here."""
import ast
import ctypes as C
import inspect
import os

import numpy as np
import pytest
import torch

from superresolution_amd import _lib, edsr_model, rcan_model, realesrgan_model, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    missing = [s for s in _lib.EXPORTS if not hasattr(L, s)]
    assert not missing, missing
    # every function the headers declare is in EXPORTS
    import re
    declared = set()
    for hname in ("srmi355.h", "edsr.h", "rcan.h", "realesrgan.h", "sr.h", "mi355_executor.h"):
        text = open(os.path.join(ROOT, "include", hname)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared |= set(re.findall(r"\b((?:srmi|edsr|rcan|realesrgan|sr)_[a-z0-9_]+)\s*\(", text))
    assert declared <= set(_lib.EXPORTS), declared - set(_lib.EXPORTS)
    assert L.srmi_version().startswith(b"srmi355")
    assert L.srmi_device_count() >= 0


def test_state_dict_layout_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "ops.npz"))
    for nm, mod in [("rrdbnet", realesrgan_model.RRDBNet()), ("edsr", edsr_model.EDSR()),
                    ("rcan", rcan_model.RCAN(scale=4))]:
        sd = mod.state_dict()
        assert list(sd.keys()) == list(g[nm + "_keys"]), nm
        assert [repr(tuple(v.shape)) for v in sd.values()] == list(g[nm + "_shapes"]), nm


def test_constructor_signatures_match_reference():
    def names(f):
        return [p for p in inspect.signature(f).parameters if p not in ("self", "compute_dtype", "trunk_f32", "device", "growth_round8")]
    assert names(realesrgan_model.RRDBNet.__init__) == ["num_in_ch", "num_out_ch", "num_feat", "num_block", "num_grow_ch", "scale"]
    assert names(edsr_model.EDSR.__init__) == ["n_resblocks", "n_feats", "scale", "rgb_range", "res_scale", "conv"]
    assert names(rcan_model.RCAN.__init__) == ["scale", "n_resgroups", "n_resblocks", "n_feats", "reduction", "res_scale", "rgb_range"]
    d = inspect.signature(edsr_model.EDSR.__init__).parameters
    assert (d["n_resblocks"].default, d["n_feats"].default, d["scale"].default, d["res_scale"].default) == (32, 256, 4, 0.1)
    with pytest.raises(ValueError, match="Unsupported scale"):
        realesrgan_model.RRDBNet(scale=3)
    with pytest.raises(NotImplementedError):
        edsr_model.EDSR(n_resblocks=1, n_feats=32, scale=5)
    # building blocks: the reference's argument order (edsr_model.py:31,52; rcan_model.py:39,81,102)
    assert names(edsr_model.ResBlock.__init__) == ["conv", "n_feats", "kernel_size", "bias", "bn", "act", "res_scale"]
    assert names(edsr_model.Upsampler.__init__) == ["conv", "scale", "n_feats", "bn", "act", "bias"]
    assert names(rcan_model.RCAB.__init__) == ["conv", "n_feats", "kernel_size", "reduction", "bias", "bn", "act", "res_scale"]
    assert names(rcan_model.ResidualGroup.__init__) == ["conv", "n_feats", "kernel_size", "reduction", "act", "res_scale", "n_resblocks"]
    conv, relu = edsr_model.default_conv, torch.nn.ReLU(True)
    rb = edsr_model.ResBlock(conv, 32, 3, act=relu, res_scale=0.1)
    assert list(rb.state_dict()) == ["body.0.weight", "body.0.bias", "body.2.weight", "body.2.bias"] and rb.res_scale == 0.1
    assert list(edsr_model.Upsampler(conv, 4, 32, act=False).state_dict()) == ["0.weight", "0.bias", "2.weight", "2.bias"]
    rg = rcan_model.ResidualGroup(conv, 32, 3, 16, relu, 1, 2)
    assert "body.1.body.3.conv_du.2.bias" in rg.state_dict() and "body.2.weight" in rg.state_dict()
    for bad in (lambda: edsr_model.ResBlock(conv, 32, 5), lambda: edsr_model.ResBlock(conv, 32, 3, bn=True),
                lambda: edsr_model.ResBlock(conv, 32, 3, bias=False), lambda: edsr_model.Upsampler(conv, 2, 32, act="prelu"),
                lambda: rcan_model.RCAB(conv, 32, 3, 16, act=torch.nn.PReLU()),
                lambda: rcan_model.RCAB(torch.nn.Conv2d, 32, 3, 16)):
        with pytest.raises(NotImplementedError):
            bad()


def test_filename_heuristics_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "ops.npz"))
    for name, ref_type, ref_scale in zip(g["names"], g["infer_model_type"], g["infer_scale"]):
        assert realesrgan_model.infer_model_type(str(name)) == ast.literal_eval(str(ref_type))
        assert rcan_model.infer_scale_from_model_name(str(name)) == int(ref_scale)
    assert edsr_model.get_meanshift_params() == ast.literal_eval(str(g["edsr_meanshift_params"]))
    assert rcan_model.get_meanshift_params() == ast.literal_eval(str(g["rcan_meanshift_params"]))


def test_meanshift_initialisation():
    m = edsr_model.EDSR(n_resblocks=1, n_feats=32)
    assert torch.equal(m.sub_mean.weight.view(3, 3), torch.eye(3))
    assert torch.allclose(m.sub_mean.bias, -255.0 * torch.tensor(synth.RGB_MEAN))
    assert torch.allclose(m.add_mean.bias, 255.0 * torch.tensor(synth.RGB_MEAN))
    assert not m.sub_mean.weight.requires_grad
    r = rcan_model.RCAN(scale=2, n_resgroups=1, n_resblocks=1)
    assert torch.equal(r.sub_mean.weight.view(3, 3), torch.eye(3))


def test_checkpoint_loaders(tmp_path):
    # Real-ESRGAN: params_ema > params; DataParallel prefix stripped; strict
    sd = {k: torch.from_numpy(v) for k, v in synth.rrdbnet_state_dict(num_block=6, seed=1).items()}
    p = str(tmp_path / "RealESRGAN_x4plus_anime_6B.pth")
    torch.save({"params": {}, "params_ema": {"module." + k: v for k, v in sd.items()}}, p)
    model, scale = realesrgan_model.load_realesrgan_from_checkpoint(p)
    assert scale == 4 and model.num_block == 6 and not model.training
    assert torch.equal(model.state_dict()["body.5.rdb3.conv5.bias"], sd["body.5.rdb3.conv5.bias"])
    bad = dict(sd)
    bad.pop("conv_last.bias")
    torch.save(bad, str(tmp_path / "RealESRGAN_x4plus_anime_6B_bad.pth"))
    with pytest.raises(RuntimeError):
        realesrgan_model.load_realesrgan_from_checkpoint(str(tmp_path / "RealESRGAN_x4plus_anime_6B_bad.pth"))
    # RCAN: 'state_dict' container, strict
    sd = {k: torch.from_numpy(v) for k, v in synth.rcan_state_dict(scale=2, seed=2).items()}
    p = str(tmp_path / "RCAN_BIX2.pt")
    torch.save({"state_dict": sd}, p)
    m = rcan_model.load_rcan_from_checkpoint(p, rcan_model.infer_scale_from_model_name(p))
    assert m.scale == 2 and torch.equal(m.state_dict()["tail.0.0.weight"], sd["tail.0.0.weight"])
    # EDSR: custom load_state_dict semantics (edsr_model.py:126-143)
    sd = {k: torch.from_numpy(v) for k, v in synth.edsr_state_dict(n_resblocks=2, n_feats=64, scale=4, seed=3).items()}
    m = edsr_model.EDSR(n_resblocks=2, n_feats=64, scale=2)          # other scale: tail shapes differ
    m.load_state_dict(sd)                                            # tail mismatches are tolerated ...
    assert torch.equal(m.state_dict()["body.1.body.2.weight"], sd["body.1.body.2.weight"])
    wrong = dict(sd)
    wrong["head.0.weight"] = torch.zeros(8, 3, 3, 3)
    with pytest.raises(RuntimeError, match="While copying the parameter named head.0.weight"):
        m.load_state_dict(wrong)                                     # ... others are not
    extra = dict(sd)
    extra["foo.weight"] = torch.zeros(1)
    with pytest.raises(KeyError, match="unexpected key"):
        m.load_state_dict(extra, strict=True)
    m.load_state_dict(extra, strict=False)
    extra2 = dict(sd)
    extra2["tail.9.weight"] = torch.zeros(1)
    m.load_state_dict(extra2, strict=True)                           # unexpected tail keys are ignored
    p = str(tmp_path / "EDSR_x4.pt")
    torch.save({"model": {k: torch.from_numpy(v) for k, v in synth.edsr_state_dict(n_resblocks=32, n_feats=256, seed=4).items()}}, p)
    full = edsr_model.load_edsr_from_checkpoint(p, scale=4)
    assert sum(x.numel() for x in full.parameters()) == 43089947


def test_core_models_share_weights():
    full = rcan_model.RCAN(scale=4, n_resgroups=1, n_resblocks=2)
    core = rcan_model.create_core_model_from_full(full, 4)
    assert "sub_mean.weight" not in core.state_dict()
    assert all(torch.equal(v, full.state_dict()[k]) for k, v in core.state_dict().items())
    fulle = edsr_model.EDSR(n_resblocks=2, n_feats=32)
    coree = edsr_model.create_core_model_from_full(fulle, 4)
    assert len(coree.state_dict()) == len(fulle.state_dict()) - 4


def _f16_bits(x):
    return np.asarray(x, np.float32).astype(np.float16).view(np.uint16)


def test_weight_packing_layout_and_rounding():
    """srmi_pack_conv3x3: f16/bf16 rounding equals numpy / torch round-to-nearest-even, and the
    image layout is [tile][stage][tap][row][2 swizzled 16-byte chunks]."""
    L = _lib.lib()
    cout, cin = 64, 64
    w = (synth.hash_uniform("pw", (cout, cin, 3, 3), 1) * 3.0).astype(np.float32)
    w[0, 0, 0, 0] = 65520.0      # rounds to inf in f16
    w[0, 1, 0, 0] = 5.9e-8       # f16 subnormal range
    w[0, 2, 0, 0] = 2.0 ** -25   # tie -> 0
    b = synth.hash_uniform("pb", (cout,), 2)
    for dt, tile in [(_lib.F16, 64), (_lib.F16, 32), (_lib.BF16, 64), (_lib.F32, 64)]:
        nb = C.c_int()
        nbytes = L.srmi_pack_conv3x3_bytes(dt, cout, cin, tile, C.byref(nb))
        assert nbytes == cout * cin * 9 * (4 if dt == _lib.F32 else 2) and nb.value == cout
        wp = np.zeros(nbytes, np.uint8)
        bp = np.zeros(nb.value, np.float32)
        assert L.srmi_pack_conv3x3(dt, cout, cin, tile, 1, w.ctypes.data, b.ctypes.data, wp.ctypes.data, bp.ctypes.data) == 0
        assert np.array_equal(bp, b)
        esz = 4 if dt == _lib.F32 else 2
        cpb, epc = 32 // esz, 16 // esz
        nstage = cin // cpb
        img = wp.reshape(cout // tile, nstage, 9, tile, 2, 16)
        rng = np.random.default_rng(0)
        for _ in range(200):
            o, ci, tap = int(rng.integers(cout)), int(rng.integers(cin)), int(rng.integers(9))
            if _ < 3:
                o, ci, tap = 0, _, 0
            t, row = divmod(o, tile)
            st = ci // cpb
            c, e = divmod(ci % cpb, epc)
            slot = c ^ (((tap * tile + row) >> 3) & 1)
            raw = img[t, st, tap, row, slot, e * esz:(e + 1) * esz].tobytes()
            v = w[o, ci, tap // 3, tap % 3]
            if dt == _lib.F32:
                assert np.frombuffer(raw, np.float32)[0] == v
            elif dt == _lib.F16:
                assert np.frombuffer(raw, np.uint16)[0] == _f16_bits(v), (o, ci, tap, v)
            else:
                want = torch.tensor([v]).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)[0]
                assert np.frombuffer(raw, np.uint16)[0] == want
    # pixel-shuffle packing permutes output channels: packed q = ij*C + c  <-  original c*r*r + ij
    cout, r = 128, 2
    w = synth.hash_uniform("pw2", (cout, 32, 3, 3), 3)
    b = np.arange(cout, dtype=np.float32)
    nb = C.c_int()
    nbytes = L.srmi_pack_conv3x3_bytes(_lib.F32, cout, 32, 64, C.byref(nb))
    wp = np.zeros(nbytes, np.uint8)
    bp = np.zeros(nb.value, np.float32)
    assert L.srmi_pack_conv3x3(_lib.F32, cout, 32, 64, r, w.ctypes.data, b.ctypes.data, wp.ctypes.data, bp.ctypes.data) == 0
    cafter = cout // (r * r)
    assert all(bp[q] == (q % cafter) * r * r + q // cafter for q in range(cout))
    # bad shapes fail loudly
    assert L.srmi_pack_conv3x3_bytes(_lib.F16, 64, 50, 64, None) == 0
    assert L.srmi_pack_conv3x3(_lib.F16, 64, 50, 64, 1, w.ctypes.data, None, wp.ctypes.data, bp.ctypes.data) == -1
    assert b"bad shape" in L.srmi_last_error()


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_winograd_weight_packing_layout():
    """srmi_pack_wino3x3 / srmi_pack_wino1d3x3 against a numpy restatement: U = G g G^T (both axes) and U = g G^T (along
    x only) per channel pair, computed in double precision, rounded once to f16, in the documented LDS images
    [cout tile][16-channel stage][position][channel half][cout row 32][8 halves]; rows beyond cout are zero."""
    L = _lib.lib()
    rng = np.random.default_rng(5)
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
    for cout, cin in ((32, 32), (48, 64), (64, 16)):
        w = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        ntile, nstage = (cout + 31) // 32, cin // 16
        wpad = np.zeros((ntile * 32, cin, 3, 3), np.float64)
        wpad[:cout] = w
        for fn_bytes, fn, npos, U in (
                (L.srmi_pack_wino3x3_bytes, L.srmi_pack_wino3x3, 16, np.einsum("ak,ockl,bl->ocab", G, wpad, G).reshape(ntile * 32, cin, 16)),
                (L.srmi_pack_wino1d3x3_bytes, L.srmi_pack_wino1d3x3, 12, np.einsum("ockl,bl->ockb", wpad, G).reshape(ntile * 32, cin, 12))):
            nb = C.c_int(0)
            fn_bytes.restype = C.c_size_t
            nbytes = fn_bytes(cout, cin, C.byref(nb))
            assert nbytes == ntile * nstage * npos * 32 * 32 and nb.value == ntile * 32
            wp = np.full(nbytes // 2, 0x7e00, np.uint16)
            bp = np.full(nb.value, np.nan, np.float32)
            assert fn(cout, cin, w.ctypes.data, b.ctypes.data, wp.ctypes.data, bp.ctypes.data) == 0
            got = wp.view(np.float16).reshape(ntile, nstage, npos, 2, 32, 8)
            # want[t][st][pos][c][row][e] = U[32 t + row][16 st + 8 c + e][pos]
            want = U.reshape(ntile, 32, nstage, 2, 8, npos).transpose(0, 2, 5, 3, 1, 4).astype(np.float32).astype(np.float16)
            assert np.array_equal(got.view(np.uint16), want.view(np.uint16))
            assert np.array_equal(bp[:cout], b) and not bp[cout:].any()
    assert L.srmi_pack_wino3x3_bytes(32, 24, None) == 0 and L.srmi_pack_wino1d3x3_bytes(32, 24, None) == 0
    wp = np.zeros(16, np.uint16)
    assert L.srmi_pack_wino3x3(32, 24, w.ctypes.data, None, wp.ctypes.data, bp.ctypes.data) == -1


def test_subpixel_upsampling_weight_packing():
    """srmi_pack_conv3x3_up2 (no GPU): the packed image, read back as four 2x2 kernels per output channel, evaluated on the
    SOURCE grid with torch reproduces conv3x3(nearest-x2 upsampling(x)) (realesrgan_model.py:115-116) up to the one
    rounding of the summed taps; image layout [cout tile][16-channel stage][tap 2x2][row][chunk swizzle][8 halves] with packed
    channel phase * cout + o, bias repeated per phase."""
    import torch.nn.functional as F
    L = _lib.lib()
    rng = np.random.default_rng(11)
    for cout, cin, tile in ((64, 32, 32), (64, 64, 64), (32, 48, 32)):
        w = (rng.standard_normal((cout, cin, 3, 3)) * 0.1).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        nb = C.c_int(0)
        nbytes = L.srmi_pack_conv3x3_up2_bytes(_lib.F16, cout, cin, tile, C.byref(nb))
        ntile, nstage = 4 * cout // tile, cin // 16
        assert nbytes == ntile * nstage * 4 * tile * 32 and nb.value == 4 * cout
        wp = np.full(nbytes // 2, 0x7e00, np.uint16)
        bp = np.full(nb.value, np.nan, np.float32)
        assert L.srmi_pack_conv3x3_up2(_lib.F16, cout, cin, tile, w.ctypes.data, b.ctypes.data, wp.ctypes.data, bp.ctypes.data) == 0
        assert np.array_equal(bp, np.tile(b, 4))
        img = wp.view(np.float16).reshape(ntile, nstage, 4, tile, 2, 8).astype(np.float32)
        # undo the 16-byte chunk swizzle: chunk c of image row r sits at c ^ ((r >> 3) & 1)
        sw = (np.arange(tile) >> 3) & 1
        for r in range(tile):
            if sw[r]:
                img[:, :, :, r] = img[:, :, :, r, ::-1].copy()
        # weff[phase][o][ci][a][b]
        weff = img.transpose(0, 3, 1, 4, 5, 2).reshape(4 * cout, cin, 2, 2).reshape(4, cout, cin, 2, 2)
        x = torch.from_numpy(rng.standard_normal((1, cin, 7, 9)).astype(np.float32))
        ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), torch.from_numpy(w), torch.from_numpy(b), padding=1)
        got = torch.empty_like(ref)
        xp = F.pad(x, (1, 1, 1, 1))
        for py in range(2):
            for px in range(2):
                # source rows y - 1 + py .. y + py, columns x - 1 + px .. x + px
                win = xp[:, :, py:py + 8, px:px + 10]
                got[:, :, py::2, px::2] = F.conv2d(win, torch.from_numpy(np.ascontiguousarray(weff[2 * py + px])), torch.from_numpy(b))
        assert float((got - ref).abs().max()) <= 2.0 ** -10 * float(ref.abs().max())
    assert L.srmi_pack_conv3x3_up2_bytes(_lib.F32, 64, 64, 32, None) == 0
    assert L.srmi_pack_conv3x3_up2_bytes(_lib.F16, 48, 64, 32, None) == 0


def test_forward_without_gpu_fails_loudly():
    m = realesrgan_model.RRDBNet(num_block=1)
    with pytest.raises(_lib.SrmiError, match="no HIP device"):
        m(torch.zeros(1, 3, 8, 8))
    d = _lib.ModelDesc()
    d.arch, d.dtype, d.scale, d.n_feats, d.n_blocks, d.grow_ch = _lib.ARCH_RRDBNET, _lib.F16, 4, 64, 1, 32
    h = C.c_void_p()
    assert _lib.lib().srmi_model_create(C.byref(d), C.byref(h)) == -1


def test_model_create_validates_arguments():
    L = _lib.lib()
    d = _lib.ModelDesc()
    h = C.c_void_p()
    d.arch, d.dtype, d.scale, d.n_feats, d.n_blocks, d.grow_ch = _lib.ARCH_RRDBNET, _lib.F16, 3, 64, 1, 32
    assert L.srmi_model_create(C.byref(d), C.byref(h)) == -1
    assert b"Unsupported scale: 3" in L.srmi_last_error()
    d.scale, d.n_feats = 4, 50
    assert L.srmi_model_create(C.byref(d), C.byref(h)) == -1
    d.arch, d.n_feats, d.scale = _lib.ARCH_EDSR, 64, 5
    assert L.srmi_model_create(C.byref(d), C.byref(h)) == -1
    assert b"not supported" in L.srmi_last_error()
    assert L.srmi_model_create(None, C.byref(h)) == -1


def test_demo_cli_usage_and_family_heuristic(capsys):
    from superresolution_amd import demo
    assert demo.main(["demo"]) == -1
    assert "Usage:" in capsys.readouterr().out
    assert demo.main(["demo", "a", "b", "c", "--model", "nope"]) == -1
    assert demo.family_from_name("/x/EDSR_x4.srmi") == "edsr"
    assert demo.family_from_name("RCAN_BIX4.srmi") == "rcan"
    assert demo.family_from_name("RealESRGAN_x4plus.srmi") == "realesrgan"


def test_weight_packing_under_host_sanitizers():
    """csrc/pack.cpp built with AddressSanitizer and UBSan (host code only, tests/cpp/pack_sanitize.cpp): every
    (dtype, cout, cin, tile, shuffle) combination the networks use packs into exactly-sized buffers without a report
    and writes every byte of the image."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "bin", "pack_sanitize")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "bin/pack_sanitize"], check=True,
                       capture_output=True)
    env = dict(os.environ, ASAN_OPTIONS="verify_asan_link_order=0:detect_leaks=0")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout + r.stderr
