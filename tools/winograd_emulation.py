"""CPU emulation of Winograd F(2x2,3x3) in half precision for the dense-block convolutions of RRDBNet
(realesrgan_model.py:25-32), at full depth (23 blocks), against the fp32 oracle.

Question it answers: if conv1..conv5 of every dense block ran as F(2x2,3x3) on the matrix cores — input tiles
transformed with packed-f16 adds (every add rounds to f16), weights transformed in f32 on the host and rounded once,
products accumulated in f32, output transform in f32 — would the network still meet the half-precision parity bound
(u8 diff <= 1 LSB, PSNR >= 50 dB)?

Rounding schedule emulated for both forms (the f16 engine's, bench configuration): every stored activation is f16,
the grown channels x1..x4 carry 8 significant bits (then f16), the trunk is f16, weights are f16 (conv1-4 and conv5's
grown columns at 8 significant bits), accumulation f32 (torch's fp32 conv).  Layers outside the dense blocks are
direct convolutions in both forms.

  python tools/winograd_emulation.py [H W [num_block]]      -> table on stdout (markdown), test infrastructure only
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import sr_oracle as O  # noqa: E402
from superresolution_amd import synth  # noqa: E402


def r16(x):
    return x.to(torch.float16).to(torch.float32)


def r8(x):      # 8 significant bits (bfloat16's), nearest even
    return x.to(torch.bfloat16).to(torch.float32)


G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)


def wino_weights(w, round_fn_cols):
    """U[a][b] = (G g G^T)[a][b] per (cout, cin), computed in f64 and rounded ONCE per element; round_fn_cols(U, cin
    index tensor) applies the storage precision per input-channel column."""
    U = torch.einsum("ai,ocij,bj->aboc", G, w.double(), G).float()   # [4][4][cout][cin]
    return round_fn_cols(U)


def wino_conv(x, U, bias, round_v, round_final=None):
    """x: [N,C,H,W] (already at its storage precision); U: [4][4][Cout][C]; returns [N,Cout,H,W] in f32.
    V = B^T d B with every add rounded by round_v (the packed-f16 adds) and the finished V by round_final (the MFMA
    operand type, when the adds themselves run wider); M = sum_c U V in f32; Y = A^T M A in f32."""
    n, c, h, w = x.shape
    he, we = (h + 1) // 2 * 2, (w + 1) // 2 * 2
    xp = F.pad(x, (1, 1 + we - w, 1, 1 + he - h))
    th, tw = he // 2, we // 2
    d = [[xp[:, :, a:a + 2 * th:2, b:b + 2 * tw:2] for b in range(4)] for a in range(4)]
    # rows: B^T d
    t = [[None] * 4 for _ in range(4)]
    for b in range(4):
        t[0][b] = round_v(d[0][b] - d[2][b])
        t[1][b] = round_v(d[1][b] + d[2][b])
        t[2][b] = round_v(d[2][b] - d[1][b])
        t[3][b] = round_v(d[1][b] - d[3][b])
    v = [[None] * 4 for _ in range(4)]
    for a in range(4):
        v[a][0] = round_v(t[a][0] - t[a][2])
        v[a][1] = round_v(t[a][1] + t[a][2])
        v[a][2] = round_v(t[a][2] - t[a][1])
        v[a][3] = round_v(t[a][1] - t[a][3])
    if round_final is not None:
        v = [[round_final(e) for e in row] for row in v]
    m = [[F.conv2d(v[a][b], U[a][b][:, :, None, None]) for b in range(4)] for a in range(4)]
    # A^T m A,  A^T = [[1,1,1,0],[0,1,-1,-1]]
    s = [[m[0][b] + m[1][b] + m[2][b] for b in range(4)], [m[1][b] - m[2][b] - m[3][b] for b in range(4)]]
    y = torch.empty((n, U.shape[2], he, we), dtype=torch.float32)
    for i in range(2):
        y[:, :, i::2, 0::2] = s[i][0] + s[i][1] + s[i][2]
        y[:, :, i::2, 1::2] = s[i][1] - s[i][2] - s[i][3]
    return y[:, :, :h, :w] + bias.view(1, -1, 1, 1)


def wino1d_weights(w, round_fn_cols):
    """F(2,3) along x: U[ky][xi] = sum_kx G[xi][kx] g[ky][kx] per (cout, cin), f64 then rounded once."""
    U = torch.einsum("xk,ocyk->yxoc", G, w.double()).float()    # [3][4][cout][cin]
    return round_fn_cols(U)


def wino1d_conv(x, U, bias, round_v):
    """x: [N,C,H,W]; U: [3][4][Cout][C].  V = B^T d along x with every add rounded by round_v; M = sum over ky and c in
    f32; Y = A^T M in f32."""
    n, c, h, w = x.shape
    we = (w + 1) // 2 * 2
    xp = F.pad(x, (1, 1 + we - w, 1, 1))
    tw = we // 2
    d = [xp[:, :, :, b:b + 2 * tw:2] for b in range(4)]            # [N,C,H+2,tw] each
    v = [round_v(d[0] - d[2]), round_v(d[1] + d[2]), round_v(d[2] - d[1]), round_v(d[1] - d[3])]
    m = []
    for xi in range(4):
        acc = None
        for ky in range(3):
            t = F.conv2d(v[xi][:, :, ky:ky + h, :], U[ky][xi][:, :, None, None])
            acc = t if acc is None else acc + t
        m.append(acc)
    y = torch.empty((n, U.shape[2], h, we), dtype=torch.float32)
    y[:, :, :, 0::2] = m[0] + m[1] + m[2]
    y[:, :, :, 1::2] = m[1] - m[2] - m[3]
    return y[:, :, :, :w] + bias.view(1, -1, 1, 1)


def rrdbnet_emulated(sd, x, num_block, wino, v_round=r16, v_final=None, u_grown_round=r8, nf=64, gc=32):
    """The f16 engine's rounding schedule; dense-block convs direct (wino=False) or F(2x2,3x3) (wino=True)."""
    lre = lambda t: F.leaky_relu(t, 0.2)
    T = lambda k: torch.from_numpy(sd[k]) if isinstance(sd[k], np.ndarray) else sd[k]

    def direct(name, inp, wround=r16):
        return F.conv2d(inp, wround(T(name + ".weight")), T(name + ".bias"), padding=1)

    def dense_conv(name, inp, is5):
        w = T(name + ".weight")
        b = T(name + ".bias")
        cin = w.shape[1]
        grown = torch.arange(cin) >= (nf if is5 else 0)            # columns stored at 8 significant bits
        if not wino:
            wr = torch.where(grown.view(1, -1, 1, 1), r8(w), r16(w))
            return F.conv2d(inp, wr, b, padding=1)
        if wino == 2:
            U = wino1d_weights(w, lambda U: torch.where(grown.view(1, 1, 1, -1), u_grown_round(U), r16(U)))
            return wino1d_conv(inp, U, b, v_round)
        U = wino_weights(w, lambda U: torch.where(grown.view(1, 1, 1, -1), u_grown_round(U), r16(U)))
        return wino_conv(inp, U, b, v_round, v_final)

    feat = r16(direct("conv_first", x, wround=lambda t: t))           # first layer runs in f32 arithmetic
    trunk = feat
    for i in range(num_block):
        rin = trunk
        for r in (1, 2, 3):
            p = f"body.{i}.rdb{r}"
            xs = [trunk]
            for j in (1, 2, 3, 4):
                xs.append(r16(r8(lre(dense_conv(f"{p}.conv{j}", torch.cat(xs, 1), False)))))
            x5 = dense_conv(f"{p}.conv5", torch.cat(xs, 1), True)
            if r < 3:
                trunk = r16(x5 * 0.2 + trunk)
            else:
                trunk = r16((x5 * 0.2 + trunk) * 0.2 + rin)
    body = r16(direct("conv_body", trunk) + feat)
    u1 = r16(lre(direct("conv_up1", O.nearest_up2(body))))
    u2 = r16(lre(direct("conv_up2", O.nearest_up2(u1))))
    hr = r16(lre(direct("conv_hr", u2)))
    return direct("conv_last", hr)


def main():
    h = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 23
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    sd = synth.rrdbnet_state_dict(num_block=nb, seed=0)
    rows = []
    for kind, img in (("smooth", synth.smooth_image(h, w, 7)), ("noise", synth.random_image(h, w, 100))):
        with torch.no_grad():
            x = O.pre_unit(img)
            ref = O.post_unit(O.rrdbnet_forward(sd, x, num_block=nb), O.ROUND_HALF_UP)[0]
            variants = (
                ("direct, f16 engine schedule (what runs today)", dict(wino=False)),
                ("F(2x2,3x3): V adds in f16, U f16 / grown columns 8 bits", dict(wino=True)),
                ("F(2x2,3x3): V adds in f16, U all f16", dict(wino=True, u_grown_round=r16)),
                ("F(2x2,3x3): V adds in f32, rounded once to f16 as the MFMA operand, U all f16",
                 dict(wino=True, u_grown_round=r16, v_round=lambda t: t, v_final=r16)),
                ("F(2,3) along x: V adds in f16, U all f16", dict(wino=2, u_grown_round=r16)),
            )
            for label, kw in variants:
                out = rrdbnet_emulated(sd, x, nb, **kw)
                got = O.post_unit(out, O.ROUND_HALF_UP)[0]
                d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
                rows.append((kind, label, int(d.max()), 100.0 * float((d > 1).mean()), O.psnr_u8(got, ref)))
                print(f"{kind:7s} {label:75s} max {d.max()} LSB  >1 LSB {100.0 * (d > 1).mean():.4f} %  PSNR {O.psnr_u8(got, ref):.2f} dB",
                      flush=True)
    print(f"\n| frame {w}x{h}, {nb} blocks | variant | max u8 diff | > 1 LSB | PSNR vs fp32 oracle |\n|---|---|---|---|---|")
    for kind, label, mx, frac, ps in rows:
        print(f"| {kind} | {label} | {mx} | {frac:.4f} % | {ps:.2f} dB |")


if __name__ == "__main__":
    main()
