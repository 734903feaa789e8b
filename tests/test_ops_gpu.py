"""Per-kernel parity: every libmmd entry point (through the C-ABI) against the oracle's primitives.

Tolerances (rel-L2 against the fp32 CPU oracle on the same inputs):
  fp32 kernels : 2e-5   (exact-fp32 MFMA / fp32 VALU; differences are summation order only)
  bf16 kernels : 1e-2   (inputs are rounded to bf16 first and the oracle sees the ROUNDED values, so the
                         budget covers bf16 output rounding + bf16 P in the attention MFMA only)
Index/window/padding errors produce O(1) errors, far above either bound.
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F_

from helpers import rel_l2
from oracle import unet_ref as uref

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def tol(dt):
    return 2e-5 if dt == torch.float32 else 1e-2


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    from mm_diffusion import ops as o
    return o


def rnd(*shape, dt=torch.float32, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*shape, generator=g) * scale
    return x.to(dt).float()          # value representable in dt, held as fp32 for the oracle


def dev(x, dt):
    return x.to(dt).cuda()


def rows_video(x):   # [N,C,F,H,W] -> [(n f h w), C]
    return x.permute(0, 2, 3, 4, 1).reshape(-1, x.shape[1]).contiguous()


def rows_audio(x):   # [N,C,L] -> [(n l), C]
    return x.permute(0, 2, 1).reshape(-1, x.shape[1]).contiguous()


def unrows_video(r, N, F, H, W):
    return r.reshape(N, F, H, W, -1).permute(0, 4, 1, 2, 3)


# --------------------------------------------------------------------------- implicit-GEMM convs
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("tile", [64, 128, 129])
@pytest.mark.parametrize("M,Cin,Cout", [(300, 64, 96), (1024, 128, 384), (77, 32, 8)])
def test_pointwise(ops, dt, tile, M, Cin, Cout):
    x, w, b, r = rnd(M, Cin, dt=dt, seed=1), rnd(Cout, Cin, dt=dt, seed=2, scale=Cin ** -0.5), rnd(Cout, seed=3), rnd(M, Cout, dt=dt, seed=4)
    y = ops.conv_gemm(dev(x, dt), dev(w, dt), b.cuda(), residual=dev(r, dt), tile=tile)
    ref = x @ w.t() + b + r
    assert rel_l2(y.float().cpu(), ref) < tol(dt)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("res", [False, True])
@pytest.mark.parametrize("M,Cin,Cout,taps", [(128 * 300 + 37, 256, 256, "1"), (128 * 130, 192, 384 + 8, "1"), (128 * 40, 96, 128, "1"),
                                            (2 * 20 * 24 * 24, 64, 128, "3x3"), (2 * 20 * 24 * 24, 96, 64, "3x3")])
def test_direct_to_lds_variants_agree(ops, dt, res, M, Cin, Cout, taps):
    """The direct-to-LDS kernel (tile 129; uniform-tap fast addressing when Cin is a multiple of one K step, generic
    otherwise; prefetched bias / residual) against the register-staged 128 tile: same K order and epilogue arithmetic,
    so bitwise equal, on ragged M / Cout edges, with a residual and with 3x3 taps."""
    tp, dims = (ops.TAPS_1, (1, 1, 1)) if taps == "1" else (ops.TAPS_SPATIAL, (40, 24, 24))
    g = torch.Generator(device="cuda").manual_seed(M + Cin)
    x = torch.randn(M, Cin, device="cuda", generator=g).to(dt)
    w = (torch.randn(Cout, Cin * len(tp), device="cuda", generator=g) * (Cin * len(tp)) ** -0.5).to(dt)
    b = torch.randn(Cout, device="cuda", generator=g)
    r = torch.randn(M, Cout, device="cuda", generator=g).to(dt) if res else None
    y0 = ops.conv_gemm(x, w, b, taps=tp, dims=dims, residual=r, tile=128)
    for _ in range(3):
        assert torch.equal(y0, ops.conv_gemm(x, w, b, taps=tp, dims=dims, residual=r, tile=129))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("res", [False, True])
@pytest.mark.parametrize("D0,H,W,Cin,Cout", [(5, 16, 32, 64, 128), (3, 8, 16, 128, 96), (2, 24, 48, 192, 256 + 8)])
def test_halo_tile_variant(ops, dt, res, D0, H, W, Cin, Cout):
    """3x3 conv with one staged halo per channel chunk (tile 130) against the direct-to-LDS kernel: chunk-major K order, so
    equal to fp32 rounding (not bitwise); patches on every border of the frame, ragged Cout, residual."""
    M = D0 * H * W
    g = torch.Generator(device="cuda").manual_seed(M + Cin)
    x = torch.randn(M, Cin, device="cuda", generator=g).to(dt)
    w = (torch.randn(Cout, Cin * 9, device="cuda", generator=g) * (Cin * 9) ** -0.5).to(dt)
    b = torch.randn(Cout, device="cuda", generator=g)
    r = torch.randn(M, Cout, device="cuda", generator=g).to(dt) if res else None
    y0 = ops.conv_gemm(x, w, b, taps=ops.TAPS_SPATIAL, dims=(D0, H, W), residual=r, tile=129)
    y1 = ops.conv_gemm(x, w, b, taps=ops.TAPS_SPATIAL, dims=(D0, H, W), residual=r, tile=130)
    assert rel_l2(y1.float().cpu(), y0.float().cpu()) < (1e-6 if dt == torch.float32 else 4e-3)


@pytest.mark.parametrize("dt", DTYPES)
def test_halo_tile_variant_temporal_form(ops, dt):
    """The temporal k=3 conv written as D = (N, F, HW), taps (0, +-1, 0): tile 130 against the (F, HW, 1) form on tile 129."""
    N, F, HW, Cin, Cout = 2, 16, 48, 128, 128
    M = N * F * HW
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(M, Cin, device="cuda", generator=g).to(dt)
    w = (torch.randn(Cout, Cin * 3, device="cuda", generator=g) * (Cin * 3) ** -0.5).to(dt)
    b = torch.randn(Cout, device="cuda", generator=g)
    y0 = ops.conv_gemm(x, w, b, taps=ops.TAPS_TEMPORAL, dims=(F, HW, 1), tile=129)
    y1 = ops.conv_gemm(x, w, b, taps=ops.TAPS_TEMPORAL_D1, dims=(N, F, HW), tile=130)
    assert rel_l2(y1.float().cpu(), y0.float().cpu()) < (1e-6 if dt == torch.float32 else 4e-3)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("tile", [64, 128, 129])
@pytest.mark.parametrize("M,Cin,Cout,taps3,res", [(256, 64, 96, False, True), (1024, 128, 128, True, False), (192, 256, 264, False, False)])
def test_gemm_epilogue_statistics(ops, dt, tile, M, Cin, Cout, taps3, res):
    """mmd_conv_gemm_stats: per (64-row record, quad of 4 columns) sum and sum of squares of the values AS STORED, written into a quad
    slice of a wider record buffer; the output itself is bitwise the plain kernel's."""
    g = torch.Generator(device="cuda").manual_seed(M + Cout)
    taps, dims = (ops.TAPS_TEMPORAL, (4, M // 4, 1)) if taps3 else (ops.TAPS_1, (1, 1, 1))
    x = torch.randn(M, Cin, device="cuda", generator=g).to(dt)
    w = (torch.randn(Cout, Cin * len(taps), device="cuda", generator=g) * (Cin * len(taps)) ** -0.5).to(dt)
    b = torch.randn(Cout, device="cuda", generator=g)
    r = torch.randn(M, Cout, device="cuda", generator=g).to(dt) if res else None
    y0 = ops.conv_gemm(x, w, b, taps=taps, dims=dims, residual=r, tile=tile)
    Q = Cout // 4                                            # one record per QUAD of channels (round 3)
    wide = torch.full((M // 64, Q + 10, 2), 7.0, device="cuda")
    y1 = ops.conv_gemm(x, w, b, taps=taps, dims=dims, residual=r, tile=tile, stats=wide[:, 6:6 + Q, :])
    assert torch.equal(y0, y1)
    yf = y1.double().view(M // 64, 64, Q, 4)
    ref = torch.stack([yf.sum((1, 3)), (yf * yf).sum((1, 3))], dim=-1)
    got = wide[:, 6:6 + Q, :].double()
    assert float((got - ref).abs().max() / ref.abs().max()) < 2e-6
    assert float((wide[:, :6] - 7).abs().max()) == 0 and float((wide[:, 6 + Q:] - 7).abs().max()) == 0


@pytest.mark.parametrize("S,Tn,C,film", [(4, 256, 128, True), (2, 1024, 384, False), (8, 64, 512, True)])
def test_gn_finalize_from_producer_statistics(ops, S, Tn, C, film):
    """mmd_gn_finalize_stats (affine from the producer GEMM's records) against mmd_gn_stats (statistics pass over the tensor)."""
    dt = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(S * Tn)
    x = (torch.randn(S * Tn, C, device="cuda", generator=g) * 1.7 + 0.6).to(dt)
    gamma, beta = torch.randn(C, device="cuda", generator=g), torch.randn(C, device="cuda", generator=g)
    fm = torch.randn(S, 2 * C, device="cuda", generator=g) * 0.3 if film else None
    geom = ops.Geom.per_sample(S, Tn)
    a0, b0 = ops.gn_stats(x, gamma, beta, geom, film=fm)
    xf = x.float().view(S * Tn // 64, 64, C // 4, 4)
    rec = torch.zeros(S * Tn // 64, C // 4 + 16, 2, device="cuda")
    rec[:, 8:8 + C // 4, 0], rec[:, 8:8 + C // 4, 1] = xf.sum((1, 3)), (xf * xf).sum((1, 3))
    a1, b1 = ops.gn_finalize_stats(rec[:, 8:8 + C // 4, :], gamma, beta, geom, film=fm)
    assert rel_l2(a1.cpu(), a0.cpu().numpy()) < 1e-5 and rel_l2(b1.cpu(), b0.cpu().numpy()) < 1e-5


def test_gn_conv1x1_statistics_feed_the_next_norm(ops):
    """Chain as in a ResBlock tail: gn_conv1x1 (+ residual) emits the statistics of its output; the next norm's affine built from them
    equals the one from a statistics pass over the stored output."""
    dt = torch.bfloat16
    S, Tn, Cin, Cout = 2, 512, 128, 128
    g = torch.Generator(device="cuda").manual_seed(99)
    x = torch.randn(S * Tn, Cin, device="cuda", generator=g).to(dt)
    w = (torch.randn(Cout, Cin, device="cuda", generator=g) * Cin ** -0.5).to(dt)
    bias = torch.randn(Cout, device="cuda", generator=g)
    r = torch.randn(S * Tn, Cout, device="cuda", generator=g).to(dt)
    gamma, beta = torch.randn(Cin, device="cuda", generator=g), torch.randn(Cin, device="cuda", generator=g)
    geom = ops.Geom.per_sample(S, Tn)
    a, b = ops.gn_stats(x, gamma, beta, geom)
    for tile in (64, 128):
        rec = torch.zeros(S * Tn // 64, Cout // 4, 2, device="cuda")
        y = ops.gn_conv1x1(x, a, b, geom, True, w, bias, residual=r, tile=tile, stats=rec)
        y0 = ops.gn_conv1x1(x, a, b, geom, True, w, bias, residual=r, tile=tile)
        assert torch.equal(y, y0)
        g2, b2 = torch.randn(Cout, device="cuda", generator=g), torch.randn(Cout, device="cuda", generator=g)
        an, bn = ops.gn_finalize_stats(rec, g2, b2, geom)
        ar, br = ops.gn_stats(y, g2, b2, geom)
        assert rel_l2(an.cpu(), ar.cpu().numpy()) < 1e-5 and rel_l2(bn.cpu(), br.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("dt", DTYPES)
def test_strided_views(ops, dt):
    """Input, residual and output as column slices of wider buffers (free skip-concat views)."""
    M, Cin, Cout = 200, 64, 64
    xb, ob, rb = rnd(M, 160, dt=dt, seed=5), torch.zeros(M, 192), rnd(M, 128, dt=dt, seed=6)
    w, b = rnd(Cout, Cin, dt=dt, seed=7, scale=0.1), rnd(Cout, seed=8)
    xd, od, rd = dev(xb, dt), dev(ob, dt), dev(rb, dt)
    ops.conv_gemm(xd[:, 96:160], dev(w, dt), b.cuda(), residual=rd[:, 64:128], out=od[:, 64:128])
    ref = xb[:, 96:160] @ w.t() + b + rb[:, 64:128]
    assert rel_l2(od[:, 64:128].float().cpu(), ref) < tol(dt)
    assert float(od[:, :64].abs().max()) == 0 and float(od[:, 128:].abs().max()) == 0


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N,F,H,W,Cin,Cout", [(2, 3, 8, 8, 64, 64), (1, 2, 5, 7, 32, 96), (1, 16, 16, 16, 128, 128)])
def test_video_conv_2d1d(ops, dt, N, F, H, W, Cin, Cout):
    x = rnd(N, Cin, F, H, W, dt=dt, seed=9)
    ws, bs = rnd(Cout, Cin, 3, 3, dt=dt, seed=10, scale=(9 * Cin) ** -0.5), rnd(Cout, seed=11)
    wt, bt = rnd(Cout, Cout, 3, dt=dt, seed=12, scale=(3 * Cout) ** -0.5), rnd(Cout, seed=13)
    sd = {"p.video_conv_spatial.weight": ws, "p.video_conv_spatial.bias": bs,
          "p.video_conv_temporal.weight": wt, "p.video_conv_temporal.bias": bt}
    y1 = ops.conv_gemm(dev(rows_video(x), dt), dev(ops.pack_conv_weight(ws, torch.float32), dt), bs.cuda(),
                       taps=ops.TAPS_SPATIAL, dims=(N * F, H, W), tile=129)     # direct-to-LDS main loop
    ref1 = F_.conv3d(x, ws[:, :, None], bs, padding=(0, 1, 1))
    assert rel_l2(unrows_video(y1.float().cpu(), N, F, H, W), ref1) < tol(dt)
    y2 = ops.conv_gemm(y1, dev(ops.pack_conv_weight(wt, torch.float32), dt), bt.cuda(), taps=ops.TAPS_TEMPORAL, dims=(F, H * W, 1))
    ref = uref.video_conv_2d1d(x, sd, "p")
    assert rel_l2(unrows_video(y2.float().cpu(), N, F, H, W), ref) < 2 * tol(dt)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("L,d", [(100, 1), (100, 4), (64, 128), (400, 512), (257, 16)])
def test_audio_conv_dilated(ops, dt, L, d):
    N, Cin, Cout = 2, 64, 96
    x = rnd(N, Cin, L, dt=dt, seed=14)
    w, b = rnd(Cout, Cin, 3, dt=dt, seed=15, scale=(3 * Cin) ** -0.5), rnd(Cout, seed=16)
    y = ops.conv_gemm(dev(rows_audio(x), dt), dev(ops.pack_conv_weight(w, torch.float32), dt), b.cuda(), taps=ops.taps_audio(d), dims=(L, 1, 1))
    y2 = ops.conv_gemm(dev(rows_audio(x), dt), dev(ops.pack_conv_weight(w, torch.float32), dt), b.cuda(), taps=ops.taps_audio(d), dims=(L, 1, 1), tile=129)
    assert torch.equal(y, y2)          # every main-loop variant accumulates in the same k order -> bitwise equal
    ref = uref.audio_conv(x, {"p.audio_conv.weight": w, "p.audio_conv.bias": b}, "p", d)
    assert rel_l2(y.float().cpu().reshape(N, L, Cout).permute(0, 2, 1), ref) < tol(dt)


# --------------------------------------------------------------------------- GroupNorm (+FiLM, +SiLU)
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("C", [64, 128, 384, 896])
def test_groupnorm_per_sample(ops, dt, C):
    N, F, H, W = 2, 3, 6, 5
    x = rnd(N, C, F, H, W, dt=dt, seed=17) * 2 + 0.7
    x = x.to(dt).float()
    g, b, film = 1 + 0.1 * rnd(C, seed=18), rnd(C, seed=19), rnd(N, 2 * C, seed=20, scale=0.3)
    geom = ops.Geom.per_sample(N, F * H * W)
    xa = dev(rows_video(x), dt)
    a_, b_ = ops.gn_stats(xa, g.cuda(), b.cuda(), geom, film=film.cuda())
    y = ops.gn_apply(xa, a_, b_, geom, act=True)
    sc, sh = film[:, :C, None, None, None], film[:, C:, None, None, None]
    ref = F_.silu(uref.group_norm(x, g, b) * (1 + sc) + sh)
    assert rel_l2(unrows_video(y.float().cpu(), N, F, H, W), ref) < tol(dt)


@pytest.mark.parametrize("dt", DTYPES)
def test_groupnorm_attention_slices(ops, dt):
    N, C, F, H, W = 2, 64, 4, 3, 5
    x = rnd(N, C, F, H, W, dt=dt, seed=21)
    g, b = 1 + 0.1 * rnd(C, seed=22), rnd(C, seed=23)
    xa = dev(rows_video(x), dt)
    # spatial: (b f) c (h w)
    geom = ops.Geom.spatial(N, F, H * W)
    y = ops.gn_apply(xa, *ops.gn_stats(xa, g.cuda(), b.cuda(), geom), geom, act=False)
    xs = x.permute(0, 2, 1, 3, 4).reshape(N * F, C, H * W)
    ref = uref.group_norm(xs, g, b).reshape(N, F, C, H, W).permute(0, 2, 1, 3, 4)
    assert rel_l2(unrows_video(y.float().cpu(), N, F, H, W), ref) < tol(dt)
    # temporal: (b h w) c f
    geom = ops.Geom.temporal(N, F, H * W)
    y = ops.gn_apply(xa, *ops.gn_stats(xa, g.cuda(), b.cuda(), geom), geom, act=False)
    xt = x.permute(0, 3, 4, 1, 2).reshape(N * H * W, C, F)
    ref = uref.group_norm(xt, g, b).reshape(N, H, W, C, F).permute(0, 3, 4, 1, 2)
    assert rel_l2(unrows_video(y.float().cpu(), N, F, H, W), ref) < tol(dt)


def test_groupnorm_large_mean_fp32(ops):
    """Statistics are accumulated in fp64 partials: a large common offset must not cancel catastrophically."""
    N, C, R = 1, 32, 5000
    x = rnd(N * R, C, seed=24) * 0.01 + 100.0
    g, b = torch.ones(C), torch.zeros(C)
    geom = ops.Geom.per_sample(N, R)
    xa = x.cuda()
    y = ops.gn_apply(xa, *ops.gn_stats(xa, g.cuda(), b.cuda(), geom), geom, act=False)
    xd = x.double().reshape(R, 32, C // 32)
    mu = xd.mean(dim=(0, 2), keepdim=True)
    var = ((xd - mu) ** 2).mean(dim=(0, 2), keepdim=True)
    ref = ((xd - mu) / torch.sqrt(var + 1e-5)).reshape(R, C)
    assert rel_l2(y.cpu(), ref) < 2e-3


# --------------------------------------------------------------------------- attention
def _qkv_rows(N, T, C, dt, seed):
    return rnd(N * T, 3 * C, dt=dt, seed=seed)


def _ref_attn(q_rows, kv_rows, heads, ch, q_idx, k_idx):
    """rows -> oracle _attend for one (batch, group)."""
    C = heads * ch
    q = q_rows[q_idx, :C].t()[None]
    k = kv_rows[k_idx, C:2 * C].t()[None]
    v = kv_rows[k_idx, 2 * C:].t()[None]
    return uref._attend(q, k, v, heads)[0].t()


@pytest.mark.parametrize("dt,impl", [(torch.float32, 0), (torch.bfloat16, 0), (torch.bfloat16, 1)])
@pytest.mark.parametrize("T,heads,ch", [(64, 4, 32), (100, 2, 64), (4, 4, 16), (1024, 1, 64), (400, 2, 128), (130, 2, 48), (70, 1, 96)])
def test_self_attention(ops, dt, impl, T, heads, ch):
    N, G, C = 2, 3, heads * ch
    qkv = _qkv_rows(N * G, T, C, dt, 25)
    out = torch.zeros(N * G * T, C, dtype=dt, device="cuda")
    ops.attn(dev(qkv, dt), dev(qkv, dt), out, heads, ch, N, G, G * T, T, G * T, T, 1, impl=impl)
    ref = torch.cat([_ref_attn(qkv, qkv, heads, ch, torch.arange(s * T, (s + 1) * T), torch.arange(s * T, (s + 1) * T))
                     for s in range(N * G)])
    assert rel_l2(out.float().cpu(), ref) < tol(dt)


@pytest.mark.parametrize("dt,impl", [(torch.float32, 0), (torch.bfloat16, 0), (torch.bfloat16, 1)])
@pytest.mark.parametrize("F,HW,L,win,shift,heads,ch", [
    (8, 16, 64, 1, 0, 2, 32), (8, 16, 64, 1, 5, 2, 32), (8, 16, 64, 4, 3, 4, 32), (8, 4, 8, 8, 0, 4, 16),
    (16, 4, 100, 4, 2, 2, 32), (16, 4, 100, 4, 12, 2, 32), (8, 4, 32, 8, 0, 2, 32), (16, 256, 1600, 4, 7, 2, 64),
])
def test_cross_attention_windows(ops, dt, impl, F, HW, L, win, shift, heads, ch):
    """RS-MMA both directions; includes wrap-around, 1 audio token/frame, L % F != 0 (remainder queries)."""
    N, C = 2, heads * ch
    apf = L // F
    vq, aq = _qkv_rows(N, F * HW, C, dt, 26), _qkv_rows(N, L, C, dt, 27)
    sh = torch.tensor([shift], dtype=torch.int32, device="cuda")
    vo = torch.zeros(N * F * HW, C, dtype=dt, device="cuda")
    ao = torch.zeros(N * L, C, dtype=dt, device="cuda")
    ops.attn(dev(vq, dt), dev(aq, dt), vo, heads, ch, N, F, F * HW, HW, L, apf, win, shift_dev=sh, impl=impl)
    ops.attn(dev(aq, dt), dev(vq, dt), ao, heads, ch, N, F, L, apf, F * HW, HW, win, shift_dev=sh, impl=impl)
    vref, aref = torch.zeros(N * F * HW, C), torch.zeros(N * L, C)
    for n in range(N):
        for i in range(F):
            a_idx = n * L + (torch.arange(win * apf) + (i + shift) * apf) % L
            qi = n * F * HW + torch.arange(i * HW, (i + 1) * HW)
            vref[qi] = _ref_attn(vq, aq, heads, ch, qi, a_idx)
            v_idx = n * F * HW + (torch.arange(win * HW) + (i + shift) * HW) % (F * HW)
            hi = L if i == F - 1 else (i + 1) * apf
            qa = n * L + torch.arange(i * apf, hi)
            aref[qa] = _ref_attn(aq, vq, heads, ch, qa, v_idx)
    assert rel_l2(vo.float().cpu(), vref) < tol(dt)
    assert rel_l2(ao.float().cpu(), aref) < tol(dt)


@pytest.mark.parametrize("impl", [2, 3])
@pytest.mark.parametrize("F,HW,L,win,shift,heads", [
    (4, 1024, 1600, 1, 2, 2),      # ds-2 geometry: 1024 x 400 and 400 x 1024 windows (ragged last sub-tile, 2 / 4 key stages, wrap-around)
    (4, 256, 400, 4, 3, 2),        # ds-4 geometry: 256 x 400 (staged, one tile per wave), 100 x 1024 (short: per-128-query kernel)
    (2, 320, 1000, 2, 1, 1),       # 320 x 1000 / 500 x 640: ragged query tiles AND ragged key stages
])
def test_cross_attention_long_windows(ops, impl, F, HW, L, win, shift, heads):
    """The long RS-MMA windows at head width 64 on both MFMA kernels (2 = per-128-query kernel, 3 = staged-window kernel; 0 picks per
    shape), against the oracle and against each other."""
    dt, ch = torch.bfloat16, 64
    N, C = 2, heads * ch
    apf = L // F
    vq, aq = _qkv_rows(N, F * HW, C, dt, 36), _qkv_rows(N, L, C, dt, 37)
    sh = torch.tensor([shift], dtype=torch.int32, device="cuda")
    outs = {}
    for im in (impl, 0):
        vo = torch.zeros(N * F * HW, C, dtype=dt, device="cuda")
        ao = torch.zeros(N * L, C, dtype=dt, device="cuda")
        ops.attn(dev(vq, dt), dev(aq, dt), vo, heads, ch, N, F, F * HW, HW, L, apf, win, shift_dev=sh, impl=im)
        ops.attn(dev(aq, dt), dev(vq, dt), ao, heads, ch, N, F, L, apf, F * HW, HW, win, shift_dev=sh, impl=im)
        outs[im] = (vo.float().cpu(), ao.float().cpu())
    vref, aref = torch.zeros(N * F * HW, C), torch.zeros(N * L, C)
    for n in range(N):
        for i in range(F):
            a_idx = n * L + (torch.arange(win * apf) + (i + shift) * apf) % L
            qi = n * F * HW + torch.arange(i * HW, (i + 1) * HW)
            vref[qi] = _ref_attn(vq, aq, heads, ch, qi, a_idx)
            v_idx = n * F * HW + (torch.arange(win * HW) + (i + shift) * HW) % (F * HW)
            hi = L if i == F - 1 else (i + 1) * apf
            qa = n * L + torch.arange(i * apf, hi)
            aref[qa] = _ref_attn(aq, vq, heads, ch, qa, v_idx)
    for im in outs:
        assert rel_l2(outs[im][0], vref) < tol(dt) and rel_l2(outs[im][1], aref) < tol(dt)
    assert rel_l2(outs[impl][0], outs[0][0]) < 3e-3 and rel_l2(outs[impl][1], outs[0][1]) < 3e-3


@pytest.mark.parametrize("T,heads", [(1024, 2), (200, 1), (520, 2)])
def test_self_attention_staged_window(ops, T, heads):
    """Spatial self-attention at head width 64 on the staged-window kernel (1, 2 and 3+ key stages; T = 520: ragged everything)."""
    dt, ch = torch.bfloat16, 64
    N, G, C = 1, 3, heads * ch
    qkv = _qkv_rows(N * G, T, C, dt, 38)
    ref = torch.cat([_ref_attn(qkv, qkv, heads, ch, torch.arange(s * T, (s + 1) * T), torch.arange(s * T, (s + 1) * T)) for s in range(N * G)])
    for impl in (3, 2):
        out = torch.zeros(N * G * T, C, dtype=dt, device="cuda")
        ops.attn(dev(qkv, dt), dev(qkv, dt), out, heads, ch, N, G, G * T, T, G * T, T, 1, impl=impl)
        assert rel_l2(out.float().cpu(), ref) < tol(dt)


@pytest.mark.parametrize("spike_key,scale", [(250, 40.0), (70, 40.0), (3, 40.0), (250, -40.0)])
def test_attention_softmax_spike(ops, spike_key, scale):
    """Online-softmax rescale path: one key dominates (a big jump of the running reference) in the first tile, in the second and late in
    the sequence, and one key far BELOW everything."""
    T, heads, ch = 300, 1, 64
    qkv = rnd(T, 3 * 64, dt=torch.bfloat16, seed=28) * 0.3
    qkv[spike_key, 64:128] = qkv[7, :64] * scale           # key aligned (or anti-aligned) with query 7
    qkv = qkv.to(torch.bfloat16).float()
    ref = _ref_attn(qkv, qkv, heads, ch, torch.arange(T), torch.arange(T))
    for impl in (2, 3, 4):      # per-128-query kernel, staged-window kernel, DMA-staged kernel
        out = torch.zeros(T, 64, dtype=torch.bfloat16, device="cuda")
        ops.attn(dev(qkv, torch.bfloat16), dev(qkv, torch.bfloat16), out, heads, ch, 1, 1, T, T, T, T, 1, impl=impl)
        assert torch.isfinite(out.float()).all()
        assert rel_l2(out.float().cpu(), ref) < 1e-2, impl
        assert rel_l2(out.float().cpu()[7], ref[7]) < 2e-2, impl            # the row that takes the jump


def test_attention_all_scores_far_below_zero(ops):
    """Every score of every row around -60 (log2 domain): a softmax whose running reference started at 0 instead of the first tile's
    maximum would underflow every P to 0 and divide by zero."""
    T, heads, ch = 200, 1, 64
    q = torch.full((T, 64), 1.0) + rnd(T, 64, seed=5) * 0.05
    k = -q * 5.0 + rnd(T, 64, seed=6) * 0.05
    v = rnd(T, 64, seed=7)
    qkv = torch.cat([q, k, v], dim=1).to(torch.bfloat16).float()
    ref = _ref_attn(qkv, qkv, heads, ch, torch.arange(T), torch.arange(T))
    for impl in (2, 4):
        out = torch.zeros(T, 64, dtype=torch.bfloat16, device="cuda")
        ops.attn(dev(qkv, torch.bfloat16), dev(qkv, torch.bfloat16), out, heads, ch, 1, 1, T, T, T, T, 1, impl=impl)
        assert torch.isfinite(out.float()).all()
        assert rel_l2(out.float().cpu(), ref) < 1e-2, impl


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("F,HW,heads,ch", [(16, 9, 4, 64), (8, 5, 4, 16), (16, 3, 4, 96), (20, 2, 2, 32), (3, 7, 4, 128), (8, 5, 4, 64), (13, 4, 2, 32), (1, 6, 2, 64)])
def test_temporal_attention(ops, dt, F, HW, heads, ch):
    N, C = 2, heads * ch
    qkv = _qkv_rows(N, F * HW, C, dt, 29)
    out = torch.zeros(N * F * HW, C, dtype=dt, device="cuda")
    ops.attn_small(dev(qkv, dt), out, C, heads, ops.Geom.temporal(N, F, HW))
    ref = torch.zeros(N * F * HW, C)
    for n in range(N):
        for p in range(HW):
            idx = n * F * HW + torch.arange(F) * HW + p
            ref[idx] = _ref_attn(qkv, qkv, heads, ch, idx, idx)
    assert rel_l2(out.float().cpu(), ref) < tol(dt)


# --------------------------------------------------------------------------- bandwidth kernels
@pytest.mark.parametrize("dt", DTYPES)
def test_resample_and_copy(ops, dt):
    N, C, F, H, W = 2, 64, 3, 8, 6
    x = rnd(N, C, F, H, W, dt=dt, seed=30)
    xa = dev(rows_video(x), dt)
    out = torch.empty(N * F * (H // 2) * (W // 2), C, dtype=dt, device="cuda")
    ops.resample(xa, out, N * F, H, W, 2, 2, 0)
    assert rel_l2(unrows_video(out.float().cpu(), N, F, H // 2, W // 2), F_.avg_pool3d(x, (1, 2, 2))) < tol(dt)
    up = torch.empty(N * F * H * 2 * W * 2, C, dtype=dt, device="cuda")
    ops.resample(xa, up, N * F, H, W, 2, 2, 1)
    ref = x.repeat_interleave(2, dim=3).repeat_interleave(2, dim=4)
    assert torch.equal(unrows_video(up.float().cpu(), N, F, 2 * H, 2 * W), ref)
    a = rnd(N, C, 40, dt=dt, seed=31)
    aa = dev(rows_audio(a), dt)
    o = torch.empty(N * 10, C, dtype=dt, device="cuda")
    ops.resample(aa, o, N, 1, 40, 1, 4, 0)
    assert rel_l2(o.float().cpu().reshape(N, 10, C).permute(0, 2, 1), F_.avg_pool1d(a, 4)) < tol(dt)
    o = torch.empty(N * 160, C, dtype=dt, device="cuda")
    ops.resample(aa, o, N, 1, 40, 1, 4, 1)
    assert torch.equal(o.float().cpu().reshape(N, 160, C).permute(0, 2, 1), a.repeat_interleave(4, dim=2))
    wide = torch.zeros(N * 40, 3 * C, dtype=dt, device="cuda")
    ops.copy2d(aa, wide[:, C:2 * C])
    assert torch.equal(wide[:, C:2 * C], aa) and float(wide[:, :C].abs().max()) == 0


def test_temb_and_linear(ops):
    dim, N = 128, 5
    W0, b0, W2, b2 = rnd(dim, dim, seed=32, scale=dim ** -0.5), rnd(dim, seed=33), rnd(dim, dim, seed=34, scale=dim ** -0.5), rnd(dim, seed=35)
    for t in (torch.tensor([0, 1, 17, 999, 500]), torch.tensor([0.0, 0.25, 131.5, 999.0, 3.0]), torch.tensor([5, 4, 3, 2, 1], dtype=torch.int32)):
        e = uref.timestep_embedding(t, dim)
        raw = F_.linear(F_.silu(F_.linear(e, W0, b0)), W2, b2)
        o_s, o_r = torch.empty(N, dim, device="cuda"), torch.empty(N, dim, device="cuda")
        ops.temb(t.cuda(), dim, W0.cuda(), b0.cuda(), W2.cuda(), b2.cuda(), o_s, o_r)
        assert rel_l2(o_r.cpu(), raw) < 2e-5 and rel_l2(o_s.cpu(), F_.silu(raw)) < 2e-5
    J = 1234
    W, b = rnd(J, dim, seed=36), rnd(J, seed=37)
    x = rnd(N, dim, seed=38)
    y = torch.empty(N, J, device="cuda")
    ops.linear(x.cuda(), W.cuda(), b.cuda(), y)
    assert rel_l2(y.cpu(), F_.linear(x, W, b)) < 2e-5


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("H,W,L,C", [(8, 6, 50, 64), (4, 8, 52, 128), (5, 12, 8, 32), (5, 32, 40, 128), (3, 64, 36, 64)])
def test_stem_and_head(ops, dt, H, W, L, C):
    """Per-pixel kernels (W, L not multiples of 4), the four-pixel strip kernels (W % 4 == 0) and, in bf16 at W % 32 == 0, the stem on
    the fp32 MFMA (fp32 x fp32 products, fp32 accumulation)."""
    N, F = 2, 3
    xv = rnd(N, F, 3, H, W, seed=39)
    ws, bs = rnd(C, 3, 3, 3, seed=40, scale=27 ** -0.5), rnd(C, seed=41)
    out = torch.empty(N * F * H * W, C, dtype=dt, device="cuda")
    ops.stem_conv(xv.cuda(), ops.pack_edge_weight(ws).cuda(), bs.cuda(), out, N, F, 3, H, W, ops.TAPS_SPATIAL)
    ref = F_.conv3d(xv.permute(0, 2, 1, 3, 4), ws[:, :, None], bs, padding=(0, 1, 1))
    assert rel_l2(unrows_video(out.float().cpu(), N, F, H, W), ref) < tol(dt)
    xa = rnd(N, 1, L, seed=42)
    wa, ba = rnd(C, 1, 3, seed=43), rnd(C, seed=44)
    out = torch.empty(N * L, C, dtype=dt, device="cuda")
    ops.stem_conv(xa.cuda(), ops.pack_edge_weight(wa).cuda(), ba.cuda(), out, N, 1, 1, 1, L, [(0, 0, -1), (0, 0, 0), (0, 0, 1)])
    assert rel_l2(out.float().cpu().reshape(N, L, C).permute(0, 2, 1), F_.conv1d(xa, wa, ba, padding=1)) < tol(dt)
    for Co in (3, 6):
        h = rnd(N, C, F, H, W, dt=dt, seed=45)
        wh, bh = rnd(Co, C, 3, 3, 3, seed=46, scale=(27 * C) ** -0.5), rnd(Co, seed=47)
        y = torch.empty(N, F, Co, H, W, device="cuda")
        ops.head_conv(dev(rows_video(h), dt), ops.pack_edge_weight(wh).cuda(), bh.cuda(), y, N, F, H, W, ops.TAPS_3D)
        ref = F_.conv3d(h, wh, bh, padding=1).permute(0, 2, 1, 3, 4)
        assert rel_l2(y.cpu(), ref) < 2e-5
    for Co in (1, 2):
        h = rnd(N, C, L, dt=dt, seed=48)
        wh, bh = rnd(Co, C, 3, seed=49), rnd(Co, seed=50)
        y = torch.empty(N, Co, L, device="cuda")
        ops.head_conv(dev(rows_audio(h), dt), ops.pack_edge_weight(wh).cuda(), bh.cuda(), y, N, 1, 1, L, [(0, 0, -1), (0, 0, 0), (0, 0, 1)])
        assert rel_l2(y.cpu(), F_.conv1d(h, wh, bh, padding=1)) < 2e-5


@pytest.mark.parametrize("learn_sigma", [False, True])
def test_ddpm_update(ops, learn_sigma):
    from oracle import diffusion_ref as dref
    S = dref.Schedule(respacing="10", learn_sigma=learn_sigma)
    N, F, C, HW = 3, 4, 3, 20
    x, noise = rnd(N, F, C, HW, seed=51), rnd(N, F, C, HW, seed=52)
    mo = rnd(N, F, 2 * C if learn_sigma else C, HW, seed=53)
    t = torch.tensor([9, 0, 4])
    mean, logvar, x0 = dref.p_mean_variance(S, mo, x, t, 2, clip=True)
    nz = (t != 0).float().reshape(-1, 1, 1, 1)
    ref = mean + nz * torch.exp(0.5 * logvar) * noise
    tab = np.stack([S.sqrt_recip_ac, S.sqrt_recipm1_ac, S.post_c1, S.post_c2,
                    np.log(np.append(S.post_var[1], S.betas[1:])), S.post_logvar_clipped, np.log(S.betas)])
    tab = torch.from_numpy(tab).float().cuda()
    out, x0o = torch.empty(N, F, C, HW, device="cuda"), torch.empty(N, F, C, HW, device="cuda")
    ops.ddpm_update(x.cuda(), mo.cuda(), noise.cuda(), out, tab, t.cuda(), F, C, HW, 1 | (4 if learn_sigma else 0), x0_out=x0o)
    assert rel_l2(out.cpu(), ref) < 1e-6 and rel_l2(x0o.cpu(), x0) < 1e-6
    xq = torch.empty_like(out)
    ops.q_sample(x.cuda(), noise.cuda(), xq, torch.from_numpy(np.stack([S.sqrt_ac, S.sqrt_1mac])).float().cuda(), t.cuda())
    assert rel_l2(xq.cpu(), dref.q_sample(S, x, t, noise)) < 1e-6


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N,C,F,H,W,Cout,film_on", [(2, 64, 4, 6, 6, 96, True), (3, 128, 2, 8, 8, 256, False), (2, 256, 1, 12, 12, 128, True)])
def test_gn_fused_into_conv1x1(ops, dt, N, C, F, H, W, Cout, film_on):
    """GroupNorm(+FiLM)+SiLU applied inside the GEMM loader == oracle GN -> SiLU -> 1x1 conv -> + residual.
    Rows per sample (144) are not a multiple of the tile, so blocks straddle two samples."""
    x = rnd(N, C, F, H, W, dt=dt, seed=60) * 1.3 + 0.2
    x = x.to(dt).float()
    g, b = 1 + 0.1 * rnd(C, seed=61), rnd(C, seed=62)
    w, bias, r = rnd(Cout, C, dt=dt, seed=63, scale=C ** -0.5), rnd(Cout, seed=64), rnd(N * F * H * W, Cout, dt=dt, seed=65)
    xa = dev(rows_video(x), dt)
    geom = ops.Geom.per_sample(N, F * H * W)
    film = rnd(N, 2 * C, seed=66, scale=0.3) if film_on else None
    xr = uref.group_norm(x, g, b)
    if film_on:
        xr = xr * (1 + film[:, :C, None, None, None]) + film[:, C:, None, None, None]
    xr = F_.silu(xr)
    assert ops.gn_fusable(geom, C, Cout)
    a_, b_ = ops.gn_stats(xa, g.cuda(), b.cuda(), geom, film=None if film is None else film.cuda())
    for tile in (64, 128):
        y = ops.gn_conv1x1(xa, a_, b_, geom, True, dev(w, dt), bias.cuda(), residual=dev(r, dt), tile=tile)
        ref = rows_video(xr) @ w.t() + bias + r
        assert rel_l2(y.float().cpu(), ref) < (2e-5 if dt == torch.float32 else 1.5e-2)


@pytest.mark.parametrize("dt,C,R", [(torch.float32, 128, 1500), (torch.float32, 1024, 700), (torch.bfloat16, 64, 5000), (torch.bfloat16, 384, 1500),
                                    (torch.bfloat16, 896, 900), (torch.bfloat16, 2048, 300), (torch.float32, 1536, 300), (torch.float32, 2048, 37),
                                    (torch.float32, 1536, 3)])
def test_groupnorm_two_stage_path(ops, dt, C, R):
    """Slices longer than one block's share take the partial + finalize route (channel counts whose vectors do not tile the
    256-thread block, the 2048-channel SR width, and fp32 rows wider than 256 16-byte vectors - the SR U-Net's 1536-channel skip
    concatenations in fp32 mode, walked in column passes - included; R = 3 / 37: the one-block route at those widths)."""
    N = 2
    x = (rnd(N * R, C, seed=67) * 1.5 - 0.3).to(dt).float()
    g, b, film = 1 + 0.1 * rnd(C, seed=68), rnd(C, seed=69), rnd(N, 2 * C, seed=70, scale=0.3)
    geom = ops.Geom.per_sample(N, R)
    xa = dev(x, dt)
    y = ops.gn_apply(xa, *ops.gn_stats(xa, g.cuda(), b.cuda(), geom, film=film.cuda()), geom, act=False)
    ref = uref.group_norm(x.reshape(N, R, C).permute(0, 2, 1), g, b) * (1 + film[:, :C, None]) + film[:, C:, None]
    assert rel_l2(y.float().cpu(), ref.permute(0, 2, 1).reshape(-1, C)) < (2e-5 if dt == torch.float32 else tol(dt))


@pytest.mark.parametrize("learn_sigma", [False, True])
def test_loss_terms(ops, learn_sigma):
    """mse / vb reductions vs the oracle's training_losses, incl. the t == 0 decoder-NLL branch."""
    from oracle import diffusion_ref as dref
    S = dref.Schedule(respacing="10", learn_sigma=learn_sigma)
    N, F, C, HW = 3, 4, 3, 50
    g = torch.Generator().manual_seed(70)
    x0 = {"video": torch.rand(N, F, C, 5, 10, generator=g) * 2 - 1, "audio": torch.rand(N, 1, 77, generator=g) * 2 - 1}
    noise = {"video": torch.randn(N, F, C, 5, 10, generator=g), "audio": torch.randn(N, 1, 77, generator=g)}
    mo = {"video": torch.randn(N, F, 2 * C if learn_sigma else C, 5, 10, generator=g), "audio": torch.randn(N, 2 if learn_sigma else 1, 77, generator=g)}
    t = torch.tensor([0, 5, 9])
    ref = dref.training_losses(S, lambda v, a, tt: (mo["video"], mo["audio"]), x0, t, noise)
    tab = np.stack([S.sqrt_recip_ac, S.sqrt_recipm1_ac, S.post_c1, S.post_c2,
                    np.log(np.append(S.post_var[1], S.betas[1:])), S.post_logvar_clipped, np.log(S.betas)])
    tab = torch.from_numpy(tab).float().cuda()
    for key, (Fk, Ck, HWk) in (("video", (F, C, HW)), ("audio", (1, 1, 77))):
        xt = dref.q_sample(S, x0[key], t, noise[key])
        mse, vb = ops.loss_terms(mo[key].cuda(), noise[key].cuda(), tab, t.cuda(), Fk, Ck, HWk, 4 if learn_sigma else 0,
                                 x0=x0[key].cuda() if learn_sigma else None, xt=xt.cuda() if learn_sigma else None)
        np.testing.assert_allclose(mse.cpu().numpy(), ref[f"mse_{key}"].numpy(), rtol=1e-5)
        if learn_sigma:
            np.testing.assert_allclose(vb.cpu().numpy(), ref[f"vb_{key}"].numpy(), rtol=2e-4, atol=1e-6)
