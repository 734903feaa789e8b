"""The fused temporal-attention block (include/mmd.h: mmd_tattn_block; mmd_tattn.hip) through the C-ABI.

One launch for  y = x + proj_out(attention over the 16 frames of a pixel(qkv(GroupNorm32(x))))  -
/root/reference/mm_diffusion/multimodal_unet.py:246-287 as used at :485-493.  Checked against (1) a float64 torch restatement of those
lines (tolerance of the bf16 path), (2) the four-launch path it replaces (gn_small, qkv GEMM, attn_small, proj_out GEMM + residual -
each pinned against torch / the oracle elsewhere; same rounding points, so close to the last bit), (3) its own statistics records
against the stored output, (4) bitwise repeatability and batch invariance."""
import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
C, HEADS, F = 256, 4, 16          # (the kernel is built for 256 channels: the ds2 level; the 384 / 512 instances were removed in round 5)


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    from mm_diffusion import ops as o
    return o


def _case(N, HW, seed, C=C):
    g = torch.Generator(device="cuda").manual_seed(seed)
    M = N * F * HW
    # per-pixel offsets and scales so that the per-(pixel, group) moments differ
    x = torch.randn(N, F, HW, C, device="cuda", generator=g)
    x = x * (0.5 + torch.rand(N, 1, HW, 1, device="cuda", generator=g)) + 0.5 * torch.randn(N, 1, HW, C, device="cuda", generator=g)
    x = x.reshape(M, C).to(BF)
    wqkv = (torch.randn(3 * C, C, device="cuda", generator=g) * C ** -0.5).to(BF)
    wproj = (torch.randn(C, C, device="cuda", generator=g) * C ** -0.5).to(BF)
    bqkv = 0.3 * torch.randn(3 * C, device="cuda", generator=g)
    bproj = 0.3 * torch.randn(C, device="cuda", generator=g)
    gamma = 1 + 0.2 * torch.randn(C, device="cuda", generator=g)
    beta = 0.2 * torch.randn(C, device="cuda", generator=g)
    return x, wqkv, wproj, bqkv, bproj, gamma, beta


def _torch_ref(x, wqkv, wproj, bqkv, bproj, gamma, beta, N, HW, eps, C=C):
    """unet:246-287 on sequences (n, pixel) x frames, float64."""
    xd = x.double().view(N, F, HW, C)
    g = xd.view(N, F, HW, 32, C // 32)
    mean = g.mean(dim=(1, 4), keepdim=True)
    var = g.var(dim=(1, 4), unbiased=False, keepdim=True)
    h = ((g - mean) / torch.sqrt(var + eps)).view(N, F, HW, C) * gamma.double() + beta.double()
    qkv = h @ wqkv.double().t() + bqkv.double()                              # [N, F, HW, 3C]
    ch = C // HEADS
    q, k, v = (t.view(N, F, HW, HEADS, ch).permute(0, 2, 3, 1, 4) for t in qkv.split(C, dim=-1))      # [N, HW, heads, F, ch]
    w = torch.softmax((q @ k.transpose(-1, -2)) * ch ** -0.5, dim=-1)
    a = (w @ v).permute(0, 3, 1, 2, 4).reshape(N, F, HW, C)
    return (xd + a @ wproj.double().t() + bproj.double()).view(N * F * HW, C)


def _four_launch(ops, x, wqkv, wproj, bqkv, bproj, gamma, beta, N, HW, C=C):
    geom = ops.Geom.temporal(N, F, HW)
    n1 = ops.gn_small(x, gamma, beta, geom, act=False)
    qkv = ops.conv_gemm(n1, wqkv, bqkv)
    att = torch.empty_like(x)
    ops.attn_small(qkv, att, C, HEADS, geom)
    return ops.conv_gemm(att, wproj, bproj, residual=x)


@pytest.mark.parametrize("N,HW,C", [(1, 16, 256), (2, 64, 256), (1, 1024, 256), (3, 48, 256), (1, 8, 256), (4, 40, 256)])
def test_tattn_block_vs_torch_and_the_four_launch_path(ops, N, HW, C):
    x, wqkv, wproj, bqkv, bproj, gamma, beta = _case(N, HW, 100 * N + HW + C, C)
    wf = ops.tattn_pack(wqkv, wproj)
    y = torch.full_like(x, float("nan"))
    ops.tattn_block(x, wf, bqkv, bproj, gamma, beta, HEADS, N, F, HW, out=y)
    assert torch.isfinite(y.float()).all()
    ref = _torch_ref(x, wqkv, wproj, bqkv, bproj, gamma, beta, N, HW, ops.GN_EPS, C)
    assert rel_l2(y.double().cpu(), ref.cpu().numpy()) < 1e-2
    y4 = _four_launch(ops, x, wqkv, wproj, bqkv, bproj, gamma, beta, N, HW, C)
    # the attention branch alone (the residual dominates y): same rounding points, different summation orders
    d_fused, d_four = y.double() - x.double(), y4.double() - x.double()
    assert rel_l2(d_fused.cpu(), d_four.cpu().numpy()) < 1.5e-2
    assert rel_l2(y.double().cpu(), y4.double().cpu().numpy()) < 4e-3
    # and the fused result is at least as close to float64 as the four-launch one (within 20 %)
    e_fused = rel_l2(d_fused.cpu(), (ref - x.double()).cpu().numpy())
    e_four = rel_l2(d_four.cpu(), (ref - x.double()).cpu().numpy())
    assert e_fused < 1.2 * e_four + 1e-3, (e_fused, e_four)


@pytest.mark.parametrize("N,HW,C", [(1, 16, 256), (2, 256, 256), (3, 40, 256)])
def test_tattn_front_stage_is_the_spatial_proj_out(ops, N, HW, C):
    """pre = (att, bias, mid): the block's input is x + att Wpre^T + bias, i.e. the spatial block's proj_out 1x1 conv + residual in
    front (unet:485-490).  Same K order, same epilogue arithmetic and the same bf16 rounding as the row-strip GEMM, then the same
    kernel: `mid` and the output are BITWISE the two-launch sequence's."""
    x, wqkv, wproj, bqkv, bproj, gamma, beta = _case(N, HW, 9 * N + HW + C, C)
    g = torch.Generator(device="cuda").manual_seed(HW)
    att = torch.randn(x.shape, device="cuda", generator=g).to(BF)
    wpre = (torch.randn(C, C, device="cuda", generator=g) * C ** -0.5).to(BF)
    bpre = 0.3 * torch.randn(C, device="cuda", generator=g)
    mid0 = ops.conv_gemm(att, wpre, bpre, residual=x)
    y0 = ops.tattn_block(mid0, ops.tattn_pack(wqkv, wproj), bqkv, bproj, gamma, beta, HEADS, N, F, HW)
    wf = ops.tattn_pack(wqkv, wproj, wpre=wpre)
    M = x.shape[0]
    rec0, rec1 = torch.zeros(M // 64, C // 4, 2, device="cuda"), torch.full((M // 64, C // 4, 2), float("nan"), device="cuda")
    ops.tattn_block(mid0, ops.tattn_pack(wqkv, wproj), bqkv, bproj, gamma, beta, HEADS, N, F, HW, stats=rec0)
    for _ in range(2):
        mid1, y1 = torch.full_like(x, float("nan")), torch.full_like(x, float("nan"))
        ops.tattn_block(x, wf, bqkv, bproj, gamma, beta, HEADS, N, F, HW, out=y1, stats=rec1, pre=(att, bpre, mid1))
        assert torch.equal(mid0.view(torch.int16), mid1.view(torch.int16))
        assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
        assert torch.equal(rec0, rec1)


@pytest.mark.parametrize("C", [256])
def test_tattn_statistics_records(ops, C):
    N, HW = 2, 64
    x, wqkv, wproj, bqkv, bproj, gamma, beta = _case(N, HW, 7 + C, C)
    wf = ops.tattn_pack(wqkv, wproj)
    M = N * F * HW
    rec = torch.full((M // 64, C // 4, 2), float("nan"), device="cuda")
    y = ops.tattn_block(x, wf, bqkv, bproj, gamma, beta, HEADS, N, F, HW, stats=rec)
    y0 = ops.tattn_block(x, wf, bqkv, bproj, gamma, beta, HEADS, N, F, HW)
    assert torch.equal(y, y0)                               # the statistics epilogue does not touch the output
    # record n * HW / 4 + (pixel >> 2): the 16 frames of 4 consecutive pixels
    v = y.double().view(N, F, HW // 4, 4, C // 4, 4)
    want = torch.stack([v.sum(dim=(1, 3, 5)), (v * v).sum(dim=(1, 3, 5))], dim=-1).view(M // 64, C // 4, 2)
    assert torch.isfinite(rec).all()
    assert (rec.double() - want).abs().max().item() <= 1e-5 * want.abs().max().item()
    # a per-sample GroupNorm finalised from them = the norm of y
    g2, b2 = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    geom = ops.Geom.per_sample(N, F * HW)
    a, b = ops.gn_finalize_stats(rec, g2, b2, geom)
    grp = y.double().view(N, F * HW, 32, C // 32)
    mean, var = grp.mean(dim=(1, 3)), grp.var(dim=(1, 3), unbiased=False)
    rstd = 1 / torch.sqrt(var + ops.GN_EPS)
    assert (a.double().view(N, 32, -1)[:, :, 0] - rstd).abs().max().item() <= 1e-4 * rstd.abs().max().item()
    assert (b.double().view(N, 32, -1)[:, :, 0] + mean * rstd).abs().max().item() <= 1e-4 * (mean * rstd).abs().max().item() + 1e-5


def test_tattn_repeatable_and_batch_invariant(ops):
    N, HW = 2, 256
    x, wqkv, wproj, bqkv, bproj, gamma, beta = _case(N, HW, 11)
    wf = ops.tattn_pack(wqkv, wproj)
    y = ops.tattn_block(x, wf, bqkv, bproj, gamma, beta, HEADS, N, F, HW).clone()
    for _ in range(5):
        assert torch.equal(y, ops.tattn_block(x, wf, bqkv, bproj, gamma, beta, HEADS, N, F, HW))
    M1 = F * HW
    for n in range(N):
        y1 = ops.tattn_block(x[n * M1:(n + 1) * M1], wf, bqkv, bproj, gamma, beta, HEADS, 1, F, HW)
        assert torch.equal(y1, y[n * M1:(n + 1) * M1])
    # a strided view of a wider buffer in and out (the engine's skip-concat buffers)
    wide_in = torch.zeros(N * M1, 384, device="cuda", dtype=BF)
    wide_in[:, 64:320] = x
    wide_out = torch.zeros(N * M1, 512, device="cuda", dtype=BF)
    ops.tattn_block(wide_in[:, 64:320], wf, bqkv, bproj, gamma, beta, HEADS, N, F, HW, out=wide_out[:, 256:])
    assert torch.equal(wide_out[:, 256:], y) and not wide_out[:, :256].any()


def test_tattn_rejects_unsupported(ops):
    H = ops.H
    x, wqkv, wproj, bqkv, bproj, gamma, beta = _case(1, 16, 3)
    wf = ops.tattn_pack(wqkv, wproj)
    with pytest.raises(H.MMDError):
        ops.tattn_block(x, wf, bqkv, bproj, gamma, beta, 8, 1, F, 16)                    # 8 heads
    with pytest.raises(H.MMDError):
        ops.tattn_block(x[:8 * 16], wf, bqkv, bproj, gamma, beta, HEADS, 1, 8, 16)       # 8 frames
    with pytest.raises(H.MMDError):
        ops.tattn_block(x, wf, bqkv, bproj, gamma, beta, HEADS, 1, F, 16, out=x)         # in place
    with pytest.raises(H.MMDError):
        ops.tattn_block(x[:, :128], wf, bqkv, bproj, gamma, beta, HEADS, 1, F, 16)       # 128 channels
    with pytest.raises(H.MMDError):
        ops.tattn_pack(wqkv[:, :128].contiguous(), wproj)
    with pytest.raises(H.MMDError):                                                       # 384 channels: the instance was removed in round 5
        x3 = torch.zeros(F * 16, 384, device="cuda", dtype=BF)
        ops.tattn_block(x3, wf, bqkv, bproj, gamma, beta, HEADS, 1, F, 16)
    with pytest.raises(H.MMDError):                                                       # HW % 8 (straight through the C-ABI)
        H.call("mmd_tattn_block", x.data_ptr(), 256, None, 0, None, 0, wf.data_ptr(), None, bqkv.data_ptr(), bproj.data_ptr(), gamma.data_ptr(),
               beta.data_ptr(), 1e-5, torch.empty_like(x).data_ptr(), 256, 1, 16, 20, 256, 4, None, 0, H.stream_handle())
    with pytest.raises(H.MMDError):                                                       # weights packed with a front stage, none given
        ops.tattn_block(x, ops.tattn_pack(wqkv, wproj, wpre=wproj), bqkv, bproj, gamma, beta, HEADS, 1, F, 16)
    with pytest.raises(H.MMDError):                                                       # the scratch aliases the input
        ops.tattn_block(x, ops.tattn_pack(wqkv, wproj, wpre=wproj), bqkv, bproj, gamma, beta, HEADS, 1, F, 16, pre=(x.clone(), bproj, x))


def test_engine_plan_uses_the_fused_temporal_attention(monkeypatch):
    """The headline architecture (batch 1) with and without MMD_TATTN_FUSED on the REFERENCE-generated full-size fixture
    (tests/golden/full_forward.npz, round 5): the default plan carries mmd_tattn_block at the ds2 level and three launches fewer per block;
    both plans sit inside the bf16 forward bound against the reference, and the fused plan is not further from it than the unfused one."""
    from helpers import flags, gold, inputs, synth_sd
    from mm_diffusion import multimodal_script_util as msu, ops as o
    g = gold("full_forward")
    outs = []
    for on in (True, False):
        fl = flags("full", use_fp16=True)
        monkeypatch.setattr(o, "_TATTN_FUSED", on)
        model, _ = msu.create_model_and_diffusion(**fl)
        model.load_state_dict(synth_sd("full"))
        model.cuda().eval()
        v, a = inputs(fl, int(g["B"]), int(g["seed"]))
        it = iter(int(s) for s in g["shifts"])
        model.shift_source = lambda lo, hi: next(it)
        with torch.no_grad():
            ov, oa = model(v.cuda(), a.cuda(), torch.from_numpy(g["t"]).cuda())
        names = [e[2] for e in next(iter(model._engines.values())).plan]
        outs.append((ov.float().cpu(), oa.float().cpu(), names.count("mmd_tattn_block"), names.count("mmd_attn_small_fwd"), len(names)))
        model.release_engines()
    fused, unfused = outs
    assert fused[2] > 0 and unfused[2] == 0 and fused[3] == unfused[3] - fused[2], [o_[2:] for o_ in outs]
    assert fused[4] <= unfused[4] - 3 * fused[2]          # (4 per block with the spatial proj_out riding along, MMD_TATTN_PRE)
    for k, ref in ((0, g["video_out"]), (1, g["audio_out"])):
        e_f, e_u = rel_l2(fused[k], ref), rel_l2(unfused[k], ref)
        print(f"fused temporal attention vs the reference fixture: fused {e_f:.3e} unfused {e_u:.3e}")
        assert e_f < 2.5e-2 and e_u < 2.5e-2 and e_f < 1.3 * e_u + 2e-3, (k, e_f, e_u)
