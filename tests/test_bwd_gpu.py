"""Backward kernels (training step) through the C-ABI against torch autograd of the oracle's fp32 primitives.

Tolerances: fp32 kernels rel-L2 <= 5e-5 (fp32 atomics order), bf16 <= 2e-2 (bf16 activations/grad tensors, fp32 accumulate).
"""
import pytest
import torch
import torch.nn.functional as F_

from helpers import rel_l2
from oracle import unet_ref as uref

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.bfloat16]


def tol(dt):
    return 5e-5 if dt == torch.float32 else 2e-2


@pytest.fixture(scope="module")
def T():
    assert torch.cuda.is_available()
    from mm_diffusion import ops, train_ops
    return ops, train_ops


def rnd(*shape, dt=torch.float32, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dt).float()


def rows_video(x):
    return x.permute(0, 2, 3, 4, 1).reshape(-1, x.shape[1]).contiguous()


def unrows_video(r, N, F, H, W):
    return r.reshape(N, F, H, W, -1).permute(0, 4, 1, 2, 3)


def leaf(x, dt):
    return x.to(dt).cuda().requires_grad_(True)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("kind", ["pointwise", "spatial", "temporal", "audio_d4", "audio_d128"])
def test_conv_backward(T, dt, kind):
    ops, tr = T
    N, F, H, W, Cin, Cout = 2, 3, 6, 5, 64, 96
    if kind.startswith("audio"):
        L, d = 100, int(kind.split("_d")[1])
        x = rnd(N, Cin, L, dt=dt, seed=1)
        w, b = rnd(Cout, Cin, 3, dt=dt, seed=2, scale=(3 * Cin) ** -0.5), rnd(Cout, seed=3)
        xr = x.permute(0, 2, 1).reshape(-1, Cin)
        taps, dims = ops.taps_audio(d), (L, 1, 1)
        ref_fn = lambda xx, ww, bb: F_.conv1d(xx, ww, bb, padding=d, dilation=d).permute(0, 2, 1).reshape(-1, Cout)
        xin = x
        to_rows = lambda g: g.permute(0, 2, 1).reshape(-1, Cin)
    else:
        x = rnd(N, Cin, F, H, W, dt=dt, seed=1)
        xr = rows_video(x)
        xin = x
        to_rows = rows_video
        if kind == "pointwise":
            w, b = rnd(Cout, Cin, 1, 1, 1, dt=dt, seed=2, scale=Cin ** -0.5), rnd(Cout, seed=3)
            taps, dims = ops.TAPS_1, (1, 1, 1)
            ref_fn = lambda xx, ww, bb: rows_video(F_.conv3d(xx, ww, bb))
        elif kind == "spatial":
            w, b = rnd(Cout, Cin, 3, 3, dt=dt, seed=2, scale=(9 * Cin) ** -0.5), rnd(Cout, seed=3)
            taps, dims = ops.TAPS_SPATIAL, (N * F, H, W)
            ref_fn = lambda xx, ww, bb: rows_video(F_.conv3d(xx, ww[:, :, None], bb, padding=(0, 1, 1)))
        else:
            w, b = rnd(Cout, Cin, 3, dt=dt, seed=2, scale=(3 * Cin) ** -0.5), rnd(Cout, seed=3)
            taps, dims = ops.TAPS_TEMPORAL, (F, H * W, 1)
            ref_fn = lambda xx, ww, bb: rows_video(F_.conv3d(xx, ww[:, :, :, None, None], bb, padding=(1, 0, 0)))
    res = rnd(xr.shape[0], Cout, dt=dt, seed=4)
    gy = rnd(xr.shape[0], Cout, dt=dt, seed=5)
    # reference
    xc, wc, bc, rc = (t.clone().requires_grad_(True) for t in (xin, w, b, res))
    (ref_fn(xc, wc, bc) + rc).backward(gy)
    # HIP
    xd, wd, bd, rd = leaf(xr, dt), leaf(w, torch.float32), leaf(b, torch.float32), leaf(res, dt)
    y = tr.conv(xd, wd, bd, taps=taps, dims=dims, residual=rd)
    y.backward(gy.to(dt).cuda())
    assert rel_l2(xd.grad.float().cpu(), to_rows(xc.grad)) < tol(dt)
    assert rel_l2(wd.grad.cpu(), wc.grad) < tol(dt)
    assert rel_l2(bd.grad.cpu(), bc.grad) < tol(dt)
    assert rel_l2(rd.grad.float().cpu(), rc.grad) < tol(dt)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("kind", ["per_sample_film", "per_sample", "spatial", "temporal"])
def test_groupnorm_backward(T, dt, kind):
    ops, tr = T
    N, C, F, H, W = 2, 64, 4, 6, 5
    x = (rnd(N, C, F, H, W, dt=dt, seed=6) * 1.5 + 0.3).to(dt).float()
    g, b = 1 + 0.1 * rnd(C, seed=7), rnd(C, seed=8)
    gy = rnd(N * F * H * W, C, dt=dt, seed=9)
    film = rnd(N, 2 * C, seed=10, scale=0.3) if kind == "per_sample_film" else None
    act = kind.startswith("per_sample")
    xc, gc, bc = (t.clone().requires_grad_(True) for t in (x, g, b))
    fc = None if film is None else film.clone().requires_grad_(True)
    if kind.startswith("per_sample"):
        geom = ops.Geom.per_sample(N, F * H * W)
        y = F_.group_norm(xc, 32, gc, bc, 1e-5)
        if fc is not None:
            y = y * (1 + fc[:, :C, None, None, None]) + fc[:, C:, None, None, None]
        y = F_.silu(y)
    elif kind == "spatial":
        geom = ops.Geom.spatial(N, F, H * W)
        y = F_.group_norm(xc.permute(0, 2, 1, 3, 4).reshape(N * F, C, H * W), 32, gc, bc, 1e-5).reshape(N, F, C, H, W).permute(0, 2, 1, 3, 4)
    else:
        geom = ops.Geom.temporal(N, F, H * W)
        y = F_.group_norm(xc.permute(0, 3, 4, 1, 2).reshape(N * H * W, C, F), 32, gc, bc, 1e-5).reshape(N, H, W, C, F).permute(0, 3, 4, 1, 2)
    rows_video(y).backward(gy)
    xd, gd, bd = leaf(rows_video(x), dt), leaf(g, torch.float32), leaf(b, torch.float32)
    fd = None if film is None else leaf(film, torch.float32)
    yd = tr.group_norm(xd, gd, bd, geom, act, film=fd)
    assert rel_l2(unrows_video(yd.detach().float().cpu(), N, F, H, W), y.detach()) < (2e-5 if dt == torch.float32 else 1e-2)
    yd.backward(gy.to(dt).cuda())
    assert rel_l2(xd.grad.float().cpu(), rows_video(xc.grad)) < tol(dt)
    assert rel_l2(gd.grad.cpu(), gc.grad) < tol(dt) and rel_l2(bd.grad.cpu(), bc.grad) < tol(dt)
    if film is not None:
        assert rel_l2(fd.grad.cpu(), fc.grad) < tol(dt)


def _attend_rows(q, k, v, heads):
    """[Tq, C], [Tk, C] -> [Tq, C] through the oracle."""
    return uref._attend(q.t()[None], k.t()[None], v.t()[None], heads)[0].t()


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("kind,T_,heads,ch", [("spatial", 70, 2, 32), ("spatial", 64, 4, 16), ("audio", 100, 2, 64), ("temporal", 8, 4, 16), ("temporal", 16, 2, 48),
                                              ("spatial", 300, 2, 64), ("audio", 200, 1, 128), ("spatial", 130, 1, 96)])
def test_self_attention_backward(T, dt, kind, T_, heads, ch):
    ops, tr = T
    C = heads * ch
    N = 2
    if kind == "temporal":
        F, HW = T_, 5
    elif kind == "spatial":
        F, HW = 3, T_
    else:
        F, HW = 1, T_
    rows = N * F * HW
    qkv = rnd(rows, 3 * C, dt=dt, seed=11)
    gy = rnd(rows, C, dt=dt, seed=12)
    qc = qkv.clone().requires_grad_(True)
    out = torch.zeros(rows, C)
    outs = []
    if kind == "temporal":
        idx_sets = [n * F * HW + torch.arange(F) * HW + p for n in range(N) for p in range(HW)]
    elif kind == "spatial":
        idx_sets = [torch.arange(s * HW, (s + 1) * HW) for s in range(N * F)]
    else:
        idx_sets = [torch.arange(n * HW, (n + 1) * HW) for n in range(N)]
    loss = 0
    for idx in idx_sets:
        o = _attend_rows(qc[idx, :C], qc[idx, C:2 * C], qc[idx, 2 * C:], heads)
        loss = loss + (o * gy[idx]).sum()
    loss.backward()
    qd = leaf(qkv, dt)
    od = tr.SelfAttnFn.apply(qd, heads, kind, N, F, HW)
    od.backward(gy.to(dt).cuda())
    assert rel_l2(qd.grad.float().cpu(), qc.grad) < tol(dt)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("F,HW,L,win,shift,heads,ch", [(8, 16, 64, 1, 5, 2, 32), (8, 16, 64, 4, 3, 2, 32), (8, 4, 8, 8, 0, 2, 16), (16, 4, 100, 4, 12, 2, 32),
                                                     (4, 256, 400, 2, 1, 2, 64), (4, 64, 403, 3, 2, 1, 128)])
def test_cross_attention_backward(T, dt, F, HW, L, win, shift, heads, ch):
    ops, tr = T
    N, C = 2, heads * ch
    apf = L // F
    vq, aq = rnd(N * F * HW, 3 * C, dt=dt, seed=13), rnd(N * L, 3 * C, dt=dt, seed=14)
    gv, ga = rnd(N * F * HW, C, dt=dt, seed=15), rnd(N * L, C, dt=dt, seed=16)
    vc, ac = vq.clone().requires_grad_(True), aq.clone().requires_grad_(True)
    loss = 0
    for n in range(N):
        for i in range(F):
            a_idx = n * L + (torch.arange(win * apf) + (i + shift) * apf) % L
            qi = n * F * HW + torch.arange(i * HW, (i + 1) * HW)
            loss = loss + (_attend_rows(vc[qi, :C], ac[a_idx, C:2 * C], ac[a_idx, 2 * C:], heads) * gv[qi]).sum()
            v_idx = n * F * HW + (torch.arange(win * HW) + (i + shift) * HW) % (F * HW)
            hi = L if i == F - 1 else (i + 1) * apf
            qa = n * L + torch.arange(i * apf, hi)
            loss = loss + (_attend_rows(ac[qa, :C], vc[v_idx, C:2 * C], vc[v_idx, 2 * C:], heads) * ga[qa]).sum()
    loss.backward()
    vd, ad = leaf(vq, dt), leaf(aq, dt)
    vo, ao = tr.CrossAttnFn.apply(vd, ad, heads, N, F, HW, L, win, shift)
    torch.autograd.backward([vo, ao], [gv.to(dt).cuda(), ga.to(dt).cuda()])
    assert rel_l2(vd.grad.float().cpu(), vc.grad) < tol(dt)
    assert rel_l2(ad.grad.float().cpu(), ac.grad) < tol(dt)


def test_small_ops_backward(T):
    ops, tr = T
    # resample
    N, C, F, H, W = 2, 64, 2, 4, 6
    x = rnd(N, C, F, H, W, seed=17)
    for mode in (0, 1):
        xc = x.clone().requires_grad_(True)
        y = F_.avg_pool3d(xc, (1, 2, 2)) if mode == 0 else xc.repeat_interleave(2, dim=3).repeat_interleave(2, dim=4)
        gy = rnd(*y.shape, seed=18)
        y.backward(gy)
        xd = leaf(rows_video(x), torch.float32)
        yd = tr.ResampleFn.apply(xd, N * F, H, W, 2, 2, mode)
        yd.backward(rows_video(gy).cuda())
        assert rel_l2(xd.grad.cpu(), rows_video(xc.grad)) < 1e-6
    # linear + silu
    xl, w, b = rnd(3, 128, seed=19), rnd(2304, 128, seed=20, scale=0.1), rnd(2304, seed=21)
    gl = rnd(3, 2304, seed=22)
    xc, wc, bc = (t.clone().requires_grad_(True) for t in (xl, w, b))
    F_.linear(F_.silu(xc), wc, bc).backward(gl)
    xd, wd, bd = (leaf(t, torch.float32) for t in (xl, w, b))
    tr.LinearFn.apply(tr.SiluFn.apply(xd), wd, bd).backward(gl.cuda())
    assert rel_l2(xd.grad.cpu(), xc.grad) < 5e-5 and rel_l2(wd.grad.cpu(), wc.grad) < 5e-5 and rel_l2(bd.grad.cpu(), bc.grad) < 5e-5
    # mse
    o, tg = rnd(3, 4, 5, 6, seed=23), rnd(3, 4, 5, 6, seed=24)
    wgt = torch.tensor([0.5, 1.0, 2.0])
    oc = o.clone().requires_grad_(True)
    l = ((tg - oc) ** 2).mean(dim=(1, 2, 3))
    (l * wgt).sum().backward()
    od = leaf(o, torch.float32)
    ld = tr.MseLossFn.apply(od, tg.cuda())
    assert rel_l2(ld.detach().cpu(), l.detach()) < 1e-6
    (ld * wgt.cuda()).sum().backward()
    assert rel_l2(od.grad.cpu(), oc.grad) < 1e-6


def test_adamw_matches_torch(T):
    ops, tr = T
    p0, g = rnd(1000, seed=25), rnd(1000, seed=26)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref], lr=1e-2, weight_decay=0.05)
    ema_ref = p0.clone()
    p, m, v, ema = p0.clone().cuda(), torch.zeros(1000).cuda(), torch.zeros(1000).cuda(), p0.clone().cuda()
    for step in (1, 2, 3):
        ref.grad = g.clone() * step
        opt.step()
        ema_ref.mul_(0.99).add_(ref.detach(), alpha=0.01)
        ops.adamw_step(p, (g * step).cuda(), m, v, ema, 1e-2, 0.9, 0.999, 1e-8, 0.05, step, ema_rate=0.99)
    assert rel_l2(p.cpu(), ref.detach()) < 1e-6 and rel_l2(ema.cpu(), ema_ref) < 1e-6
