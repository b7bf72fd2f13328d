"""Shape fuzzer: small networks of the three architectures at image sizes drawn around the kernel's tile
boundaries (rows: multiples of 8 / 12, columns: multiples of 32, each -1 / 0 / +1, plus random sizes and
batches), f32 engine against the CPU oracle at the f32 bound, f16 engine at <= 1 LSB / >= 50 dB on the u8
output.  Then, with a third argument, that many standalone convolutions (random channel counts, tile height, cout tile,
activation, residual, dtype, image size and batch) against torch's fp32 conv2d on operands rounded to the compute
type, an int8 convolution for every fourth of them against the exact integer convolution, and as many bilinear u8
resizes at random sizes against the oracle (bit-exact); the f16 convolutions with input counts in multiples of 32 also in
their two Winograd forms against the CPU emulation of those (tools/winograd_emulation.py), every f16 / bf16 case also an RGB
layer (3 outputs, 32..256 input channels, f32 and u8 results against the exact ones) and the convolution itself behind a
nearest-x2 upsampling in its sub-pixel form, and every third case a
channel-attention block in one kernel against its two-launch form (same bits).  Every network also runs a second f16
engine with an f16 trunk (RCAN: one-kernel blocks; RRDBNet: a Winograd form of the dense-block convolutions).
Prints one line per failing case and a summary; exit status 1 if anything failed.

    python tools/fuzz_shapes.py [cases] [seed] [op_cases]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F  # noqa: E402

from oracle import sr_oracle as O  # noqa: E402
from superresolution_amd import _lib, ops, synth  # noqa: E402
from tests import helpers as H  # noqa: E402

TORCH_DT = {_lib.F32: torch.float32, _lib.F16: torch.float16, _lib.BF16: torch.bfloat16}
OP_TOL = {_lib.F32: 2e-5, _lib.F16: 2.0 ** -10, _lib.BF16: 2.0 ** -7}


def fuzz_ops(rng, cases):
    """Standalone convolutions and resizes; returns the number of failures."""
    bad = 0
    for i in range(cases):
        dt = [_lib.F32, _lib.F16, _lib.BF16][int(rng.integers(0, 3))]
        cin = 32 * int(rng.integers(1, 7))
        cout = 32 * int(rng.integers(1, 5))
        cout_tile = 64 if (cout % 64 == 0 and rng.integers(0, 2)) else 32
        mb = int(rng.choice([2, 3] if cout_tile == 64 else [2, 3, 4]))
        n, h, w = int(rng.integers(1, 4)), int(rng.integers(1, 50)), int(rng.integers(1, 100))
        act = [_lib.ACT_NONE, _lib.ACT_RELU, _lib.ACT_LRELU][int(rng.integers(0, 3))]
        rq = lambda key, shape, amp=1.0: (torch.from_numpy(synth.hash_uniform(key, shape, 1000 + i)) * amp).to(TORCH_DT[dt]).float()
        x, wt = rq("x", (n, cin, h, w)), rq("w", (cout, cin, 3, 3), 0.1)
        b = torch.from_numpy(synth.hash_uniform("b", (cout,), 1000 + i))
        ref = F.conv2d(x, wt, b, padding=1)
        ref = F.relu(ref) if act == _lib.ACT_RELU else F.leaky_relu(ref, 0.2) if act == _lib.ACT_LRELU else ref
        kw = {}
        if rng.integers(0, 2):
            r1 = rq("r1", (n, cout, h, w))
            ref = ref * 0.2 + r1
            kw = dict(alpha=0.2, res1=ops.Planes.from_nchw(r1, dt))
        pc = ops.PackedConv(wt.numpy(), b.numpy(), dt, cout_tile=cout_tile)
        try:
            got = ops.conv3x3(ops.Planes.from_nchw(x, dt), pc, act=act, rows_per_wave=mb, **kw).to_nchw().cpu()
        except RuntimeError as e:   # a combination the kernel declines, with its message
            print(f"declined: dt={dt} cin={cin} cout={cout} tile={cout_tile} mb={mb}: {e}", flush=True)
            continue
        err, scale = float((got - ref).abs().max()), max(float(ref.abs().max()), 1e-6)
        if got.shape != ref.shape or err > OP_TOL[dt] * scale:
            bad += 1
            print(f"FAIL conv dt={dt} cin={cin} cout={cout} tile={cout_tile} mb={mb} n={n} h={h} w={w} act={act} "
                  f"res={bool(kw)}: {err:.3g} > {OP_TOL[dt]:.2g} * {scale:.3g}", flush=True)
        if dt == _lib.F16 and cin % 32 == 0:
            # the two Winograd forms of the same convolution against their CPU emulations (same rounding schedule: what is
            # left is f32 summation order) and, loosely, against the exact result
            from tools import winograd_emulation as W
            for algo in (_lib.ALGO_WINOGRAD, _lib.ALGO_WINOGRAD_X):
                pw = ops.PackedConv(wt.numpy(), b.numpy(), dt, winograd=algo)
                gw = ops.conv3x3(ops.Planes.from_nchw(x, dt), pw, out_dtype=_lib.F32).to_nchw().cpu()
                if algo == _lib.ALGO_WINOGRAD:
                    emu = W.wino_conv(x, W.wino_weights(wt, W.r16), b, W.r16)
                else:
                    emu = W.wino1d_conv(x, W.wino1d_weights(wt, W.r16), b, W.r16)
                plain = F.conv2d(x, wt, b, padding=1)
                e1, e2 = float((gw - emu).abs().max()), float((gw - plain).abs().max())
                sc = max(float(plain.abs().max()), 1e-6)
                if gw.shape != emu.shape or e1 > 2e-5 * sc or e2 > 5e-3 * sc:
                    bad += 1
                    print(f"FAIL winograd algo={algo} cin={cin} cout={cout} n={n} h={h} w={w}: {e1:.3g} / {e2:.3g} on {sc:.3g}", flush=True)
        if dt != _lib.F32:
            # the RGB layer (its own kernel for 64..256 channels in steps of 64, else the direct one) against the exact result
            cin3 = 32 * int(rng.integers(1, 9))
            x3, w3 = rq("x3", (n, cin3, h, w)), rq("w3", (3, cin3, 3, 3), 0.05)
            b3 = torch.from_numpy(synth.hash_uniform("b3", (3,), 1000 + i)) + 0.5
            core = F.conv2d(x3, w3, b3, padding=1)
            p3 = ops.PackedConv(w3.numpy(), b3.numpy(), dt, cout_tile=int(rng.choice([32, 64])))
            X3 = ops.Planes.from_nchw(x3, dt)
            g32 = ops.conv3x3(X3, p3, tail=dict(mode=_lib.TAIL_F32_NCHW)).cpu()
            g8 = ops.conv3x3(X3, p3, tail=dict(mode=_lib.TAIL_U8_NHWC, lo=0.0, hi=1.0, scale=255.0, round=_lib.ROUND_HALF_UP)).cpu().numpy()
            e3 = float((g32 - core).abs().max())
            d8 = np.abs(g8.astype(np.int32) - O.post_unit(core, O.ROUND_HALF_UP).astype(np.int32))
            if g32.shape != core.shape or e3 > 5e-4 * max(float(core.abs().max()), 1e-6) or d8.max() > 1:
                bad += 1
                print(f"FAIL rgb layer dt={dt} cin={cin3} n={n} h={h} w={w}: {e3:.3g}, u8 max diff {d8.max()}", flush=True)
        if dt != _lib.F32 and mb < 4:
            # the same convolution behind a nearest-x2 upsampling, in its sub-pixel form (four 2x2 convolutions on the source grid)
            hs, ws = (h + 1) // 2, (w + 1) // 2
            xs = rq("xs", (n, cin, hs, ws))
            refu = F.conv2d(F.interpolate(xs, scale_factor=2, mode="nearest"), wt, b, padding=1)
            refu = F.relu(refu) if act == _lib.ACT_RELU else F.leaky_relu(refu, 0.2) if act == _lib.ACT_LRELU else refu
            pu = ops.PackedConv(wt.numpy(), b.numpy(), dt, cout_tile=cout_tile, subpixel_up2=True)
            gu = ops.conv3x3(ops.Planes.from_nchw(xs, dt), pu, act=act, rows_per_wave=mb, up2=True).to_nchw().cpu()
            eu, su = float((gu - refu).abs().max()), max(float(refu.abs().max()), 1e-6)
            if gu.shape != refu.shape or eu > 2 * OP_TOL[dt] * su:
                bad += 1
                print(f"FAIL sub-pixel up2 dt={dt} cin={cin} cout={cout} tile={cout_tile} mb={mb} n={n} h={hs} w={ws}: {eu:.3g} on {su:.3g}", flush=True)
        if i % 3 == 0:
            # both convolutions of a channel-attention block in one kernel against the two-launch form: same bits
            xb, yb = rq("xb", (n, 64, h, w)), rq("yb", (n, 64, h, w))
            if dt != _lib.F16:
                xb, yb = xb.half().float(), yb.half().float()
            g = torch.from_numpy(synth.hash_uniform("g", (n, 64), 4000 + i) * 0.5 + 0.5).cuda()
            mk = lambda key: ops.PackedConv((synth.hash_uniform(key, (64, 64, 3, 3), 4000 + i) * 0.08).astype(np.float16).astype(np.float32),
                                            synth.hash_uniform(key + "b", (64,), 4000 + i), _lib.F16, cout_tile=64)
            p1, p2 = mk("c1"), mk("c2")
            X, Y = ops.Planes.from_nchw(xb, _lib.F16), ops.Planes.from_nchw(yb, _lib.F16)
            s1, s2 = ops.Planes(n, 64, h, w, _lib.F16, zero=True), ops.Planes(n, 64, h, w, _lib.F16, zero=True)
            t = ops.conv3x3(X, p1, act=_lib.ACT_RELU, rows_per_wave=2, in_gate=Y, gate=g, gate_alpha=0.9, in_sum=s1)
            two = ops.conv3x3(t, p2, rows_per_wave=2).to_nchw().cpu()
            one = ops.rcab_fused(X, p1, p2, in_gate=Y, gate=g, gate_alpha=0.9, in_sum=s2).to_nchw().cpu()
            if not torch.equal(one, two) or not torch.equal(s1.to_nchw().cpu(), s2.to_nchw().cpu()):
                bad += 1
                print(f"FAIL one-kernel block n={n} h={h} w={w}: max diff {float((one - two).abs().max()):.3g}", flush=True)
        if i % 4 == 0:
            # int8 x int8 -> int32 is exact: the scaled f32 output must equal the integer convolution to 1e-6
            cin8, mb8 = 32 * int(rng.integers(1, 7)), int(rng.choice([2, 3]))
            cout8 = 32 * int(rng.integers(1, 3))
            ints = lambda key, shape: np.clip(np.rint(synth.hash_uniform(key, shape, 3000 + i) * 127), -127, 127).astype(np.float32)
            xq, wq = torch.from_numpy(ints("x8", (n, cin8, h, w))), torch.from_numpy(ints("w8", (cout8, cin8, 3, 3)))
            ws = (synth.hash_uniform("ws", (cout8,), 3000 + i) * 0.5 + 1.0).astype(np.float32) * 1e-4
            b8 = synth.hash_uniform("b8", (cout8,), 3000 + i).astype(np.float32)
            ref8 = F.conv2d(xq.double(), wq.double(), None, padding=1) * torch.from_numpy(ws).double().view(1, -1, 1, 1) \
                + torch.from_numpy(b8).double().view(1, -1, 1, 1)
            got8 = ops.conv3x3(ops.Planes.from_nchw(xq, _lib.I8, qscale=1.0), ops.PackedConv(wq.numpy(), b8, _lib.I8, cout_tile=32, wscale=ws),
                               out_dtype=_lib.F32, rows_per_wave=mb8).to_nchw().cpu().double()
            err8 = float((got8 - ref8).abs().max())
            if got8.shape != ref8.shape or err8 > 1e-6 * max(float(ref8.abs().max()), 1e-6):
                bad += 1
                print(f"FAIL int8 conv cin={cin8} cout={cout8} mb={mb8} n={n} h={h} w={w}: {err8:.3g}", flush=True)
        ih, iw, oh, ow = (int(v) for v in rng.integers(1, 200, 4))
        img = synth.random_image(ih, iw, 2000 + i)
        want = O.resize_bilinear_u8(img, ow, oh)
        have = ops.resize_bilinear_u8(torch.from_numpy(img).cuda(), ow, oh).cpu().numpy()
        if not np.array_equal(have, want):
            bad += 1
            print(f"FAIL resize {iw}x{ih} -> {ow}x{oh}: max diff {np.abs(have.astype(int) - want).max()}", flush=True)
    return bad

ARCHS = [
    dict(arch="rrdbnet", num_block=1, scale=4, input="unit"),
    dict(arch="rrdbnet", num_block=1, scale=2, input="unit"),
    dict(arch="edsr", n_resblocks=2, n_feats=64, scale=4, input="raw"),
    dict(arch="edsr", n_resblocks=1, n_feats=64, scale=3, input="raw"),
    dict(arch="rcan", scale=4, n_resgroups=1, n_resblocks=2, input="raw"),
    dict(arch="rcan", scale=2, n_resgroups=2, n_resblocks=1, input="raw"),
]


def sizes(rng, count):
    edge_h = sorted({max(1, m + d) for m in (8, 12, 16, 24, 36, 48) for d in (-1, 0, 1)})
    edge_w = sorted({max(1, m + d) for m in (32, 64, 96) for d in (-1, 0, 1)})
    out = []
    for _ in range(count):
        kind = rng.integers(0, 3)
        if kind == 0:
            out.append((int(rng.choice(edge_h)), int(rng.choice(edge_w))))
        elif kind == 1:
            out.append((int(rng.integers(1, 60)), int(rng.choice(edge_w))))
        else:
            out.append((int(rng.integers(1, 60)), int(rng.integers(1, 110))))
    return out


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    bad = 0
    worst32, worst_lsb, worst_db = 0.0, 0.0, 99.0
    t0 = time.time()
    engines = {}
    for i, (h, w) in enumerate(sizes(rng, cases)):
        meta = dict(ARCHS[i % len(ARCHS)], seed=200 + i % len(ARCHS), h=h, w=w, n=int(rng.integers(1, 4)))
        key = i % len(ARCHS)
        if key not in engines:
            m32, sd = H.build_model(meta, "f32")
            m16, _ = H.build_model(meta, "f16")
            # a second f16 engine with an f16 trunk: RCAN then runs its one-kernel blocks, RRDBNet here additionally one
            # of the Winograd forms of its dense-block convolutions
            m16b, _ = H.build_model(meta, "f16", trunk_f32=False)
            if meta["arch"] == "rrdbnet":
                m16b.engine().set_winograd(1 + key % 2)
            engines[key] = (m32, m16, {k: torch.from_numpy(v) for k, v in sd.items()}, m16b)
        m32, m16, sd, m16b = engines[key]
        x = H.input_for(meta)
        want = H.oracle_forward(meta, sd, x)
        got = m32(x.cuda()).cpu()
        err = float((got - want).abs().max())
        ok32 = got.shape == want.shape and err <= H.f32_tol(meta)
        # u8 path of the f16 engine against the oracle's u8 result (half-up rounding)
        img = synth.random_image(h, w, meta["seed"], n=meta["n"])
        scale = 255.0 if meta["input"] == "unit" else 1.0
        want_u8 = torch.floor(want.clamp(0, 1 if meta["input"] == "unit" else 255) * scale + 0.5).permute(0, 2, 3, 1)
        got_u8 = m16.engine().forward_u8(torch.from_numpy(img).cuda(), _lib.ROUND_HALF_UP).cpu().float()
        d = (got_u8 - want_u8).abs()
        mse = float((d ** 2).mean())
        psnr = 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
        ok16 = got_u8.shape == want_u8.shape and float(d.max()) <= 1 and psnr >= 50.0
        got_b = m16b.engine().forward_u8(torch.from_numpy(img).cuda(), _lib.ROUND_HALF_UP).cpu().float()
        db = (got_b - want_u8).abs()
        mse_b = float((db ** 2).mean())
        ok16 = ok16 and float(db.max()) <= 1 and (mse_b == 0 or 10 * np.log10(255.0 ** 2 / mse_b) >= 50.0)
        d = torch.maximum(d, db)
        worst32 = max(worst32, err / H.f32_tol(meta))
        worst_lsb, worst_db = max(worst_lsb, float(d.max())), min(worst_db, psnr)
        if not (ok32 and ok16):
            bad += 1
            print(f"FAIL {meta}: f32 max-abs {err:.3g} (bound {H.f32_tol(meta)}), f16 max u8 diff {float(d.max())}, {psnr:.1f} dB",
                  flush=True)
        if (i + 1) % 10 == 0:
            print(f"{i + 1}/{cases} cases, {bad} failed, {time.time() - t0:.0f} s", flush=True)
    if len(sys.argv) > 3:
        op_cases = int(sys.argv[3])
        op_bad = fuzz_ops(rng, op_cases)
        print(f"fuzz_shapes: {op_cases} standalone convolutions and resizes, {op_bad} failed")
        bad += op_bad
    print(f"fuzz_shapes: {cases} cases, seed {seed}, {bad} failed; worst f32 error {worst32:.3f} of its bound, "
          f"worst f16 result {worst_lsb:.0f} LSB / {worst_db:.1f} dB")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
