"""srmi_model_forward_u8_noalloc under HIP graph capture: the call is captured into a graph on a side stream, the
graph is replayed on new input bytes, and every replay gives the bytes srmi_model_forward_u8 gives (RCAN and the
64-feature EDSR with their one-kernel blocks and the memset nodes those plans start with)."""
import ctypes as C

import pytest
import torch

from superresolution_amd import _lib, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("meta", [dict(arch="rrdbnet", num_block=2, scale=4, seed=51, input="unit"),
                                  dict(arch="rcan", scale=2, n_resgroups=1, n_resblocks=2, seed=52, input="raw"),
                                  dict(arch="edsr", scale=2, n_resblocks=3, n_feats=64, seed=53, input="raw")],
                         ids=lambda m: m["arch"])
def test_noalloc_call_can_be_captured_and_replayed(meta):
    L = _lib.lib()
    model, _ = H.build_model(meta, "f16")
    ref_model, _ = H.build_model(meta, "f16")
    eng, ref_eng = model.engine(), ref_model.engine()
    n, h, w, s = 2, 40, 56, eng.scale
    img = torch.zeros((n, h, w, 3), dtype=torch.uint8, device="cuda")
    out = torch.zeros((n, h * s, w * s, 3), dtype=torch.uint8, device="cuda")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = L.srmi_model_forward_u8_noalloc(eng._h, img.data_ptr(), out.data_ptr(), n, h, w, _lib.ROUND_HALF_UP, stream)
    assert rc == 0, L.srmi_last_error().decode()
    for seed in (1, 2, 3):
        img.copy_(torch.from_numpy(synth.random_image(h, w, seed, n=n)))
        out.fill_(7)
        g.replay()
        torch.cuda.synchronize()
        want = ref_eng.forward_u8(img.clone(), _lib.ROUND_HALF_UP)
        assert torch.equal(out, want), f"replay with seed {seed} differs"
    assert eng.workspace_bytes() == 0


def test_regular_forward_under_capture_after_an_eager_call():
    """srmi_model_forward_u8 itself under capture: after an eager warm-up call (which recorded the model's ordering
    event outside any capture) a batch of three frames (three internal lanes that fork from and join the captured
    stream) is captured, replayed on new bytes, followed by another eager call; all give the eager bytes.  A captured
    call that would have to allocate (a larger shape than any seen) fails with a message naming srmi_model_reserve
    instead of allocating inside the capture."""
    L = _lib.lib()
    meta = dict(arch="rrdbnet", num_block=2, scale=4, seed=51, input="unit")
    model, _ = H.build_model(meta, "f16")
    ref_model, _ = H.build_model(meta, "f16")
    eng, ref_eng = model.engine(), ref_model.engine()
    n, h, w, s = 3, 40, 56, eng.scale
    img = torch.from_numpy(synth.random_image(h, w, 9, n=n)).cuda()
    out = torch.zeros((n, h * s, w * s, 3), dtype=torch.uint8, device="cuda")
    eng.forward_u8(img, _lib.ROUND_HALF_UP, out=out)          # eager: allocates workspace and lanes, records the event
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = L.srmi_model_forward_u8(eng._h, img.data_ptr(), out.data_ptr(), n, h, w, _lib.ROUND_HALF_UP, stream)
        msg = L.srmi_last_error().decode() if rc else ""
        big = torch.zeros((n, 2 * h, 2 * w, 3), dtype=torch.uint8, device="cuda")
        big_out = torch.zeros((n, 2 * h * s, 2 * w * s, 3), dtype=torch.uint8, device="cuda")
        rc_big = L.srmi_model_forward_u8(eng._h, big.data_ptr(), big_out.data_ptr(), n, 2 * h, 2 * w, _lib.ROUND_HALF_UP, stream)
        msg_big = L.srmi_last_error().decode() if rc_big else ""
    assert rc == 0, msg
    assert rc_big == -1 and "srmi_model_reserve" in msg_big
    for seed in (1, 2):
        img.copy_(torch.from_numpy(synth.random_image(h, w, seed, n=n)))
        out.fill_(7)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref_eng.forward_u8(img.clone(), _lib.ROUND_HALF_UP)), f"replay with seed {seed} differs"
    img.copy_(torch.from_numpy(synth.random_image(h, w, 5, n=n)))
    got = eng.forward_u8(img, _lib.ROUND_HALF_UP)              # eager again after the capture
    assert torch.equal(got, ref_eng.forward_u8(img.clone(), _lib.ROUND_HALF_UP))
