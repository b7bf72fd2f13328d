"""srmi_model_forward_u8_noalloc (include/srmi355.h): the allocation-free variant of srmi_model_forward_u8.
It must give the same bytes as srmi_model_forward_u8 on two calls in a row, the second on a smaller input
(the static workspace keeps the first call's contents), must leave the model's own workspace untouched
and must reject inputs whose workspace exceeds the static array."""
import ctypes as C

import numpy as np
import pytest
import torch

from superresolution_amd import _lib, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu

CASES = [
    dict(arch="rrdbnet", num_block=2, scale=4, seed=51, input="unit"),
    dict(arch="rcan", scale=2, n_resgroups=1, n_resblocks=2, seed=52, input="raw"),
    dict(arch="edsr", n_resblocks=2, n_feats=64, scale=3, seed=53, input="raw"),
]


def _call(fn, eng, img, out):
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    n, h, w, _ = img.shape
    return fn(eng._h, img.data_ptr(), out.data_ptr(), n, h, w, _lib.ROUND_HALF_UP, stream)


@pytest.mark.parametrize("meta", CASES, ids=lambda m: m["arch"])
def test_noalloc_matches_forward_u8_on_two_calls_in_a_row(meta):
    L = _lib.lib()
    model, _ = H.build_model(meta, "f16")
    ref_model, _ = H.build_model(meta, "f16")
    eng, ref_eng = model.engine(), ref_model.engine()
    assert eng.workspace_bytes() == 0
    s = eng.scale
    # second call: fewer images of a smaller size
    for n, h, w, seed in ((2, 40, 56, 1), (1, 23, 31, 2)):
        img = torch.from_numpy(synth.random_image(h, w, seed, n=n)).cuda()
        want = ref_eng.forward_u8(img, _lib.ROUND_HALF_UP)
        got = torch.full((n, h * s, w * s, 3), 7, dtype=torch.uint8, device="cuda")
        rc = _call(L.srmi_model_forward_u8_noalloc, eng, img, got)
        assert rc == 0, L.srmi_last_error().decode()
        torch.cuda.synchronize()
        assert torch.equal(got, want), f"{meta['arch']} {n}x{h}x{w}: noalloc output differs from srmi_model_forward_u8"
    # the model never allocated a workspace of its own
    assert eng.workspace_bytes() == 0
    assert ref_eng.workspace_bytes() > 0


def test_noalloc_rejects_inputs_beyond_the_static_workspace():
    L = _lib.lib()
    assert L.srmi_static_workspace_bytes() == 64 << 20
    meta = CASES[0]
    model, _ = H.build_model(meta, "f16")
    eng = model.engine()
    img = torch.zeros((1, 339, 510, 3), dtype=torch.uint8, device="cuda")
    out = torch.zeros((1, 1356, 2040, 3), dtype=torch.uint8, device="cuda")
    assert _call(L.srmi_model_forward_u8_noalloc, eng, img, out) == -1
    msg = L.srmi_last_error().decode()
    assert "static workspace" in msg and str(64 << 20) in msg
    assert eng.workspace_bytes() == 0
    # the regular entry point still serves that size on the same model
    assert _call(L.srmi_model_forward_u8, eng, img, out) == 0
    torch.cuda.synchronize()
