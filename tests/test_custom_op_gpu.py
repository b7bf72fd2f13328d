"""The forward passes as registered PyTorch operators (superresolution_amd/custom_ops.py): dispatcher-visible, with a
fake-tensor shape function, usable inside torch.compile(fullgraph=True), same bytes as the direct C-ABI call.  The
reference calls `model(img)` under torch.no_grad() (realesrgan/mtk/python/test/test_pytorch.py:127-128)."""
import ctypes as C

import numpy as np
import pytest
import torch

from superresolution_amd import _lib, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _model():
    meta = dict(arch="rrdbnet", num_block=2, scale=4, seed=71, input="unit")
    model, _ = H.build_model(meta, "f16")
    return model


def test_operators_are_registered_and_match_the_c_abi():
    model = _model()
    eng = model.engine()
    assert hasattr(torch.ops.srmi355, "forward_u8") and hasattr(torch.ops.srmi355, "forward_f32")
    img = torch.from_numpy(synth.random_image(24, 40, 72, n=2)).cuda()
    got = torch.ops.srmi355.forward_u8(eng.handle(), eng.scale, img, _lib.ROUND_HALF_UP)
    want = torch.empty((2, 96, 160, 3), dtype=torch.uint8, device="cuda")
    L = _lib.lib()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.srmi_model_forward_u8(eng._h, img.data_ptr(), want.data_ptr(), 2, 24, 40, _lib.ROUND_HALF_UP, stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    x = img.permute(0, 3, 1, 2).float() / 255.0
    y = torch.ops.srmi355.forward_f32(eng.handle(), eng.scale, x)
    assert y.shape == (2, 3, 96, 160) and torch.equal(y, eng.forward_f32(x, out=torch.empty_like(y)))
    # shape function on meta / fake tensors: no kernel runs
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        f = torch.empty((3, 17, 29, 3), dtype=torch.uint8, device="cuda")
        o = torch.ops.srmi355.forward_u8(eng.handle(), 4, f, 1)
        assert o.shape == (3, 68, 116, 3) and o.dtype == torch.uint8
    torch.library.opcheck(torch.ops.srmi355.forward_u8.default, (eng.handle(), eng.scale, img, _lib.ROUND_HALF_UP),
                          test_utils=("test_schema", "test_faketensor"))


def test_model_call_inside_torch_compile_fullgraph():
    """`model(x)` as the reference calls it, traced by torch.compile(fullgraph=True) together with the torch ops around
    it (u8 -> float in, clamp / round -> u8 out, realesrgan/mtk/python/test/test_pytorch.py:18-84): one graph, no
    break, the same bytes as eager."""
    model = _model()
    model.engine()                       # weights uploaded outside the traced region

    def pipeline(img_u8):
        x = img_u8.permute(0, 3, 1, 2).to(torch.float32) * (1.0 / 255.0)
        y = model(x)
        return (y.clamp(0.0, 1.0) * 255.0).round().to(torch.uint8).permute(0, 2, 3, 1)

    img = torch.from_numpy(synth.random_image(24, 40, 73, n=2)).cuda()
    with torch.no_grad():
        eager = pipeline(img)
        compiled = torch.compile(pipeline, fullgraph=True, backend="aot_eager")
        got = compiled(img)
        torch.cuda.synchronize()
        assert got.shape == (2, 96, 160, 3) and torch.equal(got, eager)
        # another batch size re-traces through the shape function (or runs dynamically) and still matches
        img2 = torch.from_numpy(synth.random_image(24, 40, 74, n=3)).cuda()
        assert torch.equal(compiled(img2), pipeline(img2))
