"""srmi_model_forward_u8_noalloc under HIP graph capture: the call is captured into a graph on a side stream, the
graph is replayed on new input bytes, and every replay gives the bytes srmi_model_forward_u8 gives."""
import ctypes as C

import pytest
import torch

from superresolution_amd import _lib, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("meta", [dict(arch="rrdbnet", num_block=2, scale=4, seed=51, input="unit"),
                                  dict(arch="rcan", scale=2, n_resgroups=1, n_resblocks=2, seed=52, input="raw")],
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
