"""GroupNorm finalised in the CONSUMER (include/mmd.h: mmd_gn_rec; mmd_gn_apply_rec, mmd_gn_conv1x1_rec) through the C-ABI.

The consumer kernels turn the producers' quad records of their slice into the fused affine in their prologue.  Reference here: the
two-launch path they replace (mmd_gn_finalize_stats + mmd_gn_apply / mmd_gn_conv1x1 on the strip tile), which the other GPU tests pin
against torch and the oracle.  The sums are fp32 records added in double - exact for these inputs - and the arithmetic after them is
the same, so the outputs are compared BITWISE; a float64 torch GroupNorm of the same tensor bounds both."""
import pytest
import torch

from mm_diffusion import ops as _ops

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(_ops._STRIP_MODE != "pin", reason="needs the row-strip kernel (MMD_GEMM_STRIP=pin)")]
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    from mm_diffusion import ops as o
    return o


def _records(x):
    """Quad records of the stored values, as the GEMM epilogues leave them: [rows / 64, C / 4, 2] fp32 (sum, sum of squares)."""
    M, C = x.shape
    v = x.float().view(M // 64, 64, C // 4, 4)
    return torch.stack([v.sum(dim=(1, 3)), (v * v).sum(dim=(1, 3))], dim=-1).contiguous()


def _case(S, Tn, C, film, seed, ld_extra=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    # a per-(slice, group) offset and scale: the moments differ between slices and groups
    x = torch.randn(S, Tn, 32, C // 32, device="cuda", generator=g)
    x = x * (0.5 + torch.rand(S, 1, 32, 1, device="cuda", generator=g)) + torch.randn(S, 1, 32, 1, device="cuda", generator=g)
    x = x.reshape(S * Tn, C).to(BF)
    rec_full = torch.zeros(S * Tn // 64, C // 4 + ld_extra, 2, device="cuda")
    rec = rec_full[:, ld_extra // 2: ld_extra // 2 + C // 4, :]
    rec.copy_(_records(x))
    gamma = 1 + 0.2 * torch.randn(C, device="cuda", generator=g)
    beta = 0.2 * torch.randn(C, device="cuda", generator=g)
    fl = 0.3 * torch.randn(S, 2 * C + 64, device="cuda", generator=g)[:, :2 * C] if film else None
    return x, rec, gamma, beta, fl


def _torch_gn(x, S, Tn, C, gamma, beta, film, act):
    v = x.double().view(S, Tn, 32, C // 32)
    mean = v.mean(dim=(1, 3), keepdim=True)
    var = v.var(dim=(1, 3), unbiased=False, keepdim=True)
    y = ((v - mean) / torch.sqrt(var + _ops.GN_EPS)).view(S, Tn, C) * gamma.double() + beta.double()
    if film is not None:
        y = y * (1 + film[:, None, :C].double()) + film[:, None, C:].double()
    if act:
        y = y * torch.sigmoid(y)
    return y.view(S * Tn, C)


@pytest.mark.parametrize("S,Tn,C,film,act", [
    (4, 1024, 512, True, True),          # ds8 ResBlock in-norm: 16 records per slice
    (4, 4096, 384, True, True),          # ds4: 64 records x 96 quads (three passes of the moments loop)
    (64, 64, 512, False, False),         # ds8 spatial-attention norm: ONE record per slice
    (64, 256, 384, False, False),
    (3, 192, 128, False, True),          # odd slice count, 3 records
    (2, 1600 * 64 // 64, 896, True, True),   # a skip concat: 7 quads per group
    (5, 128, 1024, True, False),
])
def test_gn_apply_rec_is_the_two_launch_path(ops, S, Tn, C, film, act):
    x, rec, gamma, beta, fl = _case(S, Tn, C, film, S * 1000 + Tn + C, ld_extra=8)
    geom = ops.Geom.per_sample(S, Tn)
    a, b = ops.gn_finalize_stats(rec, gamma, beta, geom, film=fl)
    y0 = ops.gn_apply(x, a, b, geom, act=act)
    src = ops.RecAffine(rec, gamma, beta, fl)
    for _ in range(2):
        y1 = torch.full_like(y0, float("nan"))
        ops.gn_apply_rec(x, src, geom, act=act, out=y1)
        assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
    ref = _torch_gn(x, S, Tn, C, gamma, beta, fl, act)
    assert (y1.double() - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("S,Tn,K,N,film,act,res,stats", [
    (4, 1024, 512, 512, True, True, True, True),         # ds8 out-conv: K = 512, one fragment per wave
    (4, 1024, 512, 1536, False, False, False, False),    # ds8 qkv
    (4, 4096, 384, 384, True, True, True, True),         # ds4 out-conv
    (64, 256, 384, 1152, False, False, False, False),    # ds4 spatial-attention qkv: 4 records per slice, two slices per 256... (BR = 128: one)
    (64, 1024, 256, 768, False, False, False, False),    # ds2 spatial-attention qkv (BR = 256)
    (6, 320, 256, 256, True, True, True, True),          # strips that cross a slice boundary (320 = 1.25 x 256)
    (5, 192, 384, 96, False, True, False, True),         # ... with one fragment per wave (192 = 1.5 x 128)
    (3, 256, 128, 128, True, False, True, False),        # K = 128
])
def test_gn_conv1x1_rec_is_the_two_launch_path(ops, S, Tn, K, N, film, act, res, stats):
    x, rec, gamma, beta, fl = _case(S, Tn, K, film, S * 77 + Tn + K + N)
    M = S * Tn
    g = torch.Generator(device="cuda").manual_seed(N)
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(BF)
    bias = torch.randn(N, device="cuda", generator=g)
    r = torch.randn(M, N, device="cuda", generator=g).to(BF) if res else None
    geom = ops.Geom.per_sample(S, Tn)
    a, b = ops.gn_finalize_stats(rec, gamma, beta, geom, film=fl)
    st0 = torch.zeros(M // 64, N // 4, 2, device="cuda") if stats else None
    y0 = ops.gn_conv1x1(x, a, b, geom, act, w, bias, residual=r, tile=131, stats=st0)
    src = ops.RecAffine(rec, gamma, beta, fl)
    for _ in range(2):
        st1 = torch.full_like(st0, float("nan")) if stats else None
        y1 = torch.full_like(y0, float("nan"))
        ops.gn_conv1x1_rec(x, src, geom, act, w, bias, residual=r, out=y1, stats=st1)
        assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
        if stats:
            assert torch.equal(st0, st1)
    n = _torch_gn(x, S, Tn, K, gamma, beta, fl, act).to(BF).double()
    ref = n @ w.double().t() + bias.double() + (r.double() if res else 0)
    assert (y1.double() - ref).abs().max().item() <= 3e-2 * max(1.0, ref.abs().max().item())


def test_rec_consumers_reject_what_they_cannot_take(ops):
    x, rec, gamma, beta, _ = _case(2, 256, 256, False, 5)
    src = ops.RecAffine(rec, gamma, beta)
    H = ops.H
    with pytest.raises(H.MMDError):
        ops.gn_apply_rec(x, src, ops.Geom.per_sample(16, 32))                       # slices that are not whole records
    with pytest.raises(H.MMDError):
        ops.gn_apply_rec(x[:, :128], ops.RecAffine(rec, gamma, beta), ops.Geom.per_sample(2, 256))      # channel mismatch
    w = torch.randn(64, 256, device="cuda").to(BF)
    with pytest.raises(H.MMDError):
        ops.gn_conv1x1_rec(x, src, ops.Geom.per_sample(4, 128), False, w, None)    # slices shorter than one strip (256 rows at K = 256)
    with pytest.raises(H.MMDError):
        ops.RecAffine(rec[:, :24, :], gamma[:96], beta[:96])                       # 96 channels: groups are not whole quads


def test_engine_plan_uses_the_consumer_finalize(monkeypatch):
    """The launch plan of the mid-size model with and without MMD_GN_REC: the default plan carries the consumer-side launches and
    fewer mmd_gn_finalize_stats; the outputs agree (to the last bit on most runs - the double sums are exact - but whole plans are
    only promised to bf16 rounding noise: the two plans may autotune different tiles for a producer)."""
    from helpers import flags, inputs, rel_l2
    from mm_diffusion import multimodal_script_util as msu, ops as o
    from mm_diffusion.synth import synth_init_
    fl = flags("mid", use_fp16=True)
    outs = []
    for on in (True, False):
        monkeypatch.setattr(o, "_GN_REC", on)
        model, _ = msu.create_model_and_diffusion(**fl)
        synth_init_(model)
        model.cuda().eval()
        v, a = inputs(fl, 2, 3)
        import random
        random.seed(5)
        with torch.no_grad():
            ov, oa = model(v.cuda(), a.cuda(), torch.tensor([17, 400]).cuda())
        names = [e[2] for e in next(iter(model._engines.values())).plan]
        outs.append((ov.clone(), oa.clone(), names.count("mmd_gn_finalize_stats"), names.count("mmd_gn_apply_rec") + names.count("mmd_gn_conv1x1_rec")))
        model.release_engines()
    assert outs[0][3] > 0 and outs[1][3] == 0 and outs[0][2] < outs[1][2], [o_[2:] for o_ in outs]
    assert rel_l2(outs[0][0].cpu(), outs[1][0].cpu().numpy()) < 5e-3 and rel_l2(outs[0][1].cpu(), outs[1][1].cpu().numpy()) < 5e-3
