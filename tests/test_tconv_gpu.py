"""The temporal k = 3 conv of VideoConv with stationary activations (include/mmd.h: mmd_tconv; mmd_tconv.hip) through the C-ABI.

/root/reference/mm_diffusion/multimodal_unet.py:83-99 (video_conv_temporal: Conv1d over the frames of every pixel, zero padding).
The kernel keeps the K order (tap-major, then channel) and the epilogue arithmetic of mmd_conv_gemm, so its OUTPUT is compared bitwise
with conv_gemm on the temporal taps - which the other suites pin against torch and the oracle - and, independently, against
torch.nn.functional.conv1d in float64; its statistics records (own row order) against the stored output."""
import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
F = 16


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    from mm_diffusion import ops as o
    return o


def _case(N, HW, Cin, Cout, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(N * F * HW, Cin, device="cuda", generator=g).to(BF)
    w = (torch.randn(Cout, 3 * Cin, device="cuda", generator=g) * (3 * Cin) ** -0.5).to(BF)        # K = tap * Cin + ci
    b = torch.randn(Cout, device="cuda", generator=g)
    return x, w, b


@pytest.mark.parametrize("N,HW,Cin,Cout", [
    (4, 1024, 256, 256),        # ds2, batch 4 (headline)
    (2, 256, 384, 384),         # ds4
    (3, 64, 512, 512),          # ds8
    (1, 8, 256, 128),           # one workgroup; fewer output channels
    (1, 24, 512, 256), (2, 40, 384, 192),
])
def test_tconv_is_bitwise_conv_gemm_and_matches_torch(ops, N, HW, Cin, Cout):
    x, w, b = _case(N, HW, Cin, Cout, N + HW + Cin + Cout)
    y0 = ops.conv_gemm(x, w, b, taps=ops.TAPS_TEMPORAL, dims=(F, HW, 1), tile=129)
    wf = ops.tconv_pack(w)
    for _ in range(2):
        y1 = torch.full_like(y0, float("nan"))
        ops.tconv(x, wf, b, Cout, N, F, HW, out=y1)
        assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
    y2 = ops.tconv(x, wf, None, Cout, N, F, HW)                           # no bias
    assert torch.equal(y2, ops.conv_gemm(x, w, None, taps=ops.TAPS_TEMPORAL, dims=(F, HW, 1), tile=129))
    if N * HW <= 1024:
        xt = x.double().view(N, F, HW, Cin).permute(0, 2, 3, 1).reshape(N * HW, Cin, F)
        wt = w.double().view(Cout, 3, Cin).permute(0, 2, 1)               # [Cout, Cin, tap]
        ref = torch.nn.functional.conv1d(xt, wt, b.double(), padding=1).view(N, HW, Cout, F).permute(0, 3, 1, 2).reshape(N * F * HW, Cout)
        assert rel_l2(y1.double().cpu(), ref.cpu().numpy()) < 5e-3


def test_tconv_statistics_records(ops):
    N, HW, Cin, Cout = 2, 64, 256, 256
    x, w, b = _case(N, HW, Cin, Cout, 5)
    wf = ops.tconv_pack(w)
    M = N * F * HW
    rec = torch.full((M // 64, Cout // 4, 2), float("nan"), device="cuda")
    y = ops.tconv(x, wf, b, Cout, N, F, HW, stats=rec)
    assert torch.equal(y, ops.tconv(x, wf, b, Cout, N, F, HW))
    v = y.double().view(N, F, HW // 4, 4, Cout // 4, 4)                   # record n * HW / 4 + (pixel >> 2): 16 frames of 4 pixels
    want = torch.stack([v.sum(dim=(1, 3, 5)), (v * v).sum(dim=(1, 3, 5))], dim=-1).view(M // 64, Cout // 4, 2)
    assert torch.isfinite(rec).all()
    assert (rec.double() - want).abs().max().item() <= 1e-5 * want.abs().max().item()
    for Cin2, Cout2 in ((384, 384), (512, 512)):                          # 32-channel chunks: one sub-tile per chunk
        x2, w2, b2 = _case(1, 32, Cin2, Cout2, Cin2)
        rec2 = torch.full((F * 32 // 64, Cout2 // 4, 2), float("nan"), device="cuda")
        y2 = ops.tconv(x2, ops.tconv_pack(w2), b2, Cout2, 1, F, 32, stats=rec2)
        v2 = y2.double().view(1, F, 8, 4, Cout2 // 4, 4)
        want2 = torch.stack([v2.sum(dim=(1, 3, 5)), (v2 * v2).sum(dim=(1, 3, 5))], dim=-1).view(-1, Cout2 // 4, 2)
        assert (rec2.double() - want2).abs().max().item() <= 1e-5 * want2.abs().max().item()


def test_tconv_batch_invariant_and_strided_views(ops):
    N, HW, Cin, Cout = 2, 128, 256, 256
    x, w, b = _case(N, HW, Cin, Cout, 9)
    wf = ops.tconv_pack(w)
    y = ops.tconv(x, wf, b, Cout, N, F, HW).clone()
    M1 = F * HW
    for n in range(N):
        assert torch.equal(ops.tconv(x[n * M1:(n + 1) * M1], wf, b, Cout, 1, F, HW), y[n * M1:(n + 1) * M1])
    wide_in = torch.zeros(N * M1, 512, device="cuda", dtype=BF)
    wide_in[:, 128:384] = x
    wide_out = torch.zeros(N * M1, 384, device="cuda", dtype=BF)
    ops.tconv(wide_in[:, 128:384], wf, b, Cout, N, F, HW, out=wide_out[:, 128:])
    assert torch.equal(wide_out[:, 128:], y) and not wide_out[:, :128].any()


def test_tconv_rejects_unsupported(ops):
    H = ops.H
    x, w, b = _case(1, 8, 256, 256, 1)
    wf = ops.tconv_pack(w)
    with pytest.raises(H.MMDError):
        ops.tconv(x[:, :128], wf, b, 256, 1, F, 8)                         # 128 input channels
    with pytest.raises(H.MMDError):
        ops.tconv(x[:8 * 8], wf, b, 256, 1, 8, 8)                          # 8 frames
    with pytest.raises(H.MMDError):
        ops.tconv(x, wf, b, 128, 1, F, 8)                                  # weight image for another Cout
    with pytest.raises(H.MMDError):
        ops.tconv(x, wf, b, 256, 1, F, 8, out=x)                           # in place
    with pytest.raises(H.MMDError):
        ops.tconv_pack(w[:, :300].contiguous())
    with pytest.raises(H.MMDError):                                        # HW % 8 (straight through the C-ABI)
        H.call("mmd_tconv", x.data_ptr(), 256, wf.data_ptr(), None, torch.empty_like(x).data_ptr(), 256, 1, 16, 12, 256, 256, None, 0, H.stream_handle())


def test_engine_plan_uses_tconv(monkeypatch):
    """The headline architecture's plan (batch 1) with and without MMD_TCONV: the default plan carries mmd_tconv for the ResBlock
    temporal convs at 256 + channels.  The op is bitwise the GEMM it replaces, but its records are in its own order, so norms over
    slices smaller than a sample fall back to the statistics pass and whole plans differ by bf16 noise: each is measured against
    the fp32 engine, and the default plan must not be further from it."""
    from helpers import flags, inputs
    from mm_diffusion import multimodal_script_util as msu, ops as o
    from mm_diffusion.synth import synth_init_
    import random
    outs = []
    for fp16, on in ((False, True), (True, True), (True, False)):
        fl = flags("full", use_fp16=fp16)
        monkeypatch.setattr(o, "_TCONV", on)
        model, _ = msu.create_model_and_diffusion(**fl)
        synth_init_(model)
        model.cuda().eval()
        v, a = inputs(fl, 1, 3)
        random.seed(5)
        with torch.no_grad():
            ov, oa = model(v.cuda(), a.cuda(), torch.tensor([417]).cuda())
        names = [e[2] for e in next(iter(model._engines.values())).plan]
        outs.append((ov.float().cpu(), oa.float().cpu(), names.count("mmd_tconv")))
        model.release_engines()
    ref, on_, off_ = outs
    assert ref[2] == 0 and on_[2] > 0 and off_[2] == 0, [o_[2] for o_ in outs]
    for k in (0, 1):
        e_on, e_off = rel_l2(on_[k], ref[k].numpy()), rel_l2(off_[k], ref[k].numpy())
        assert e_on < 1.3 * e_off + 2e-3 and e_on < 3e-2, (k, e_on, e_off)
