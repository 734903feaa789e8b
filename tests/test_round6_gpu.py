"""Round-6 kernels (through the C ABI).

attn_pipe_kernel<64> (mmd_attn_fwd impl 5): the software-pipelined flash attention at head width 64 whose loop iteration is one
hand-written asm statement (tools/gen_attn_pipe.py).  Same arithmetic in the same order as attn_mfma_kernel (impl 2, the kernel
tests/test_ops_gpu.py pins against the oracle's attention, unet:221-240, 507-564): the outputs must be BITWISE equal - on circular
windows with a shift, key counts that are not multiples of 64 (ragged last tile, zero-filled DMA rows), 1 .. 17 key tiles (prologue /
odd and even pipelined iterations / drain), ragged query tiles, waves without queries, the last group's remainder, and on inputs that
force the online-softmax rescale."""
import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    from mm_diffusion import ops as o
    return o


CASES = [
    ("spatial 1024", 2, 4, 4 * 1024, 1024, 4 * 1024, 1024, 1, 4), ("v<-a", 2, 16, 16 * 256, 256, 1600, 100, 1, 4),
    ("a<-v window 4", 1, 16, 1600, 100, 16 * 256, 256, 4, 6), ("ragged keys / queries", 2, 8, 8 * 77, 77, 8 * 50, 50, 3, 2),
    ("last group takes the remainder", 1, 16, 1610, 100, 16 * 64, 64, 8, 2), ("one short tile", 1, 16, 16 * 64, 64, 16 * 25, 25, 1, 8),
    ("one full tile", 1, 4, 4 * 200, 200, 4 * 64, 64, 1, 2), ("two tiles", 1, 4, 4 * 130, 130, 4 * 128, 128, 1, 2),
    ("three tiles, ragged", 1, 4, 4 * 130, 130, 4 * 150, 150, 1, 2), ("four tiles", 2, 2, 2 * 300, 300, 2 * 256, 256, 1, 1),
    ("v<-a ds2 full size", 1, 16, 16 * 1024, 1024, 6400, 400, 1, 4), ("a<-v ds2 full size (16-query tail block)", 1, 16, 6400, 400, 16 * 1024, 1024, 1, 4),
    ("600 queries, 7 tiles", 2, 4, 4 * 600, 600, 4 * 440, 440, 1, 2), ("17 tiles", 1, 2, 2 * 160, 160, 2 * 1050, 1050, 1, 2),
    ("window wraps around the end of the key rows", 1, 8, 8 * 128, 128, 8 * 96, 96, 5, 2)]


@pytest.mark.parametrize("name,N,F,qr,qg,kr,kg,win,heads", CASES)
def test_attn_pipe_kernel_is_bitwise_the_mfma_kernel(ops, name, N, F, qr, qg, kr, kg, win, heads):
    ch = 64
    C = heads * ch
    g = torch.Generator(device="cuda").manual_seed(qr + kr)
    q = torch.randn(N * qr, 3 * C, device="cuda", generator=g).to(BF)
    kv = torch.randn(N * kr, 3 * C, device="cuda", generator=g).to(BF)
    for shift in (0, 5):
        sh = torch.tensor([shift], dtype=torch.int32, device="cuda")
        o2 = torch.zeros(N * qr, C, device="cuda", dtype=BF)
        o5 = torch.full((N * qr, C), 7.0, device="cuda", dtype=BF)
        ops.attn(q, kv, o2, heads, ch, N, F, qr, qg, kr, kg, win, shift_dev=sh, impl=2)
        ops.attn(q, kv, o5, heads, ch, N, F, qr, qg, kr, kg, win, shift_dev=sh, impl=5)
        torch.cuda.synchronize()
        assert torch.equal(o2, o5), f"{name} shift {shift}: rel-L2 {rel_l2(o5.float().cpu(), o2.float().cpu().numpy()):.3e}"


@pytest.mark.parametrize("spike_key,scale", [(250, 40.0), (70, 40.0), (3, 40.0), (250, -40.0), (135, 25.0)])
def test_attn_pipe_kernel_softmax_spike(ops, spike_key, scale):
    """The running maximum jumps in the first tile, in a pipelined iteration, in the last (ragged) tile; a key far below everything."""
    T, heads, ch = 300, 1, 64
    g = torch.Generator().manual_seed(28)
    qkv = (torch.randn(T, 3 * 64, generator=g) * 0.3).to(BF).float()
    qkv[spike_key, 64:128] = qkv[7, :64] * scale           # key aligned (or anti-aligned) with query 7
    qkv = qkv.to(BF).cuda()
    outs = {}
    for impl in (2, 5):
        out = torch.zeros(T, 64, dtype=BF, device="cuda")
        ops.attn(qkv, qkv, out, heads, ch, 1, 1, T, T, T, T, 1, impl=impl)
        outs[impl] = out
    torch.cuda.synchronize()
    assert torch.isfinite(outs[5].float()).all()
    assert torch.equal(outs[2], outs[5])


def test_attn_pipe_kernel_scores_far_below_zero(ops):
    """Every score around -60 (log2 domain): the first tile's rescale from the -1e30 start must not underflow the sums."""
    T, heads, ch = 200, 1, 64
    g = torch.Generator().manual_seed(5)
    q = torch.full((T, 64), 1.0) + torch.randn(T, 64, generator=g) * 0.05
    k = -q * 5.0 + torch.randn(T, 64, generator=g) * 0.05
    v = torch.randn(T, 64, generator=g)
    qkv = torch.cat([q, k, v], dim=1).to(BF).cuda()
    outs = {}
    for impl in (2, 5):
        out = torch.zeros(T, 64, dtype=BF, device="cuda")
        ops.attn(qkv, qkv, out, heads, ch, 1, 1, T, T, T, T, 1, impl=impl)
        outs[impl] = out
    torch.cuda.synchronize()
    assert torch.isfinite(outs[5].float()).all()
    assert torch.equal(outs[2], outs[5])


def test_attn_pipe_kernel_repeatable_under_load(ops):
    """200 launches of the ds2 cross-attention shape on two streams at once (the kernel shares its SIMDs with another instance of
    itself in a different phase): every output bitwise the first - the hand-counted waits hold under contention."""
    N, F, qr, qg, kr, kg, win, heads, ch = 2, 16, 16 * 1024, 1024, 6400, 400, 1, 4, 64
    C = heads * ch
    g = torch.Generator(device="cuda").manual_seed(3)
    q = torch.randn(N * qr, 3 * C, device="cuda", generator=g).to(BF)
    kv = torch.randn(N * kr, 3 * C, device="cuda", generator=g).to(BF)
    ref = torch.zeros(N * qr, C, device="cuda", dtype=BF)
    ops.attn(q, kv, ref, heads, ch, N, F, qr, qg, kr, kg, win, impl=2)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = [torch.zeros_like(ref) for _ in range(4)]
    bad = 0
    for rep in range(50):
        for o in outs:
            o.zero_()
        torch.cuda.synchronize()
        for i, o in enumerate(outs):
            if i % 2:
                with torch.cuda.stream(side):
                    ops.attn(q, kv, o, heads, ch, N, F, qr, qg, kr, kg, win, impl=5)
            else:
                ops.attn(q, kv, o, heads, ch, N, F, qr, qg, kr, kg, win, impl=5)
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(o, ref) else 1 for o in outs)
    assert bad == 0, f"{bad} of 200 launches differ"


# --------------------------------------------------------------------------- audio in_layers in one launch (mmd_aconv)
ACONV_CASES = [   # N, L, Cin, Cout, dilation
    (2, 400, 512, 512, 2), (4, 400, 1024, 512, 1), (2, 400, 512, 512, 512), (2, 1600, 384, 384, 256), (2, 1600, 896, 384, 512),
    (1, 6400, 256, 256, 32), (2, 6400, 128, 256, 16), (1, 25600, 128, 128, 4), (1, 256, 128, 128, 8), (3, 448, 640, 256, 64)]


@pytest.mark.parametrize("N,L,Cin,Cout,dil", ACONV_CASES)
def test_aconv_is_bitwise_gn_apply_plus_conv_gemm(ops, N, L, Cin, Cout, dil):
    """mmd_aconv (GroupNorm + SiLU + dilated k = 3 AudioConv in one launch; unet:339-346, 108-131, nn.py:16-33) against the two launches
    it replaces (mmd_gn_apply, mmd_conv_gemm with the taps (-d, 0, d)): same K order and rounding points -> bitwise equal; against fp32
    torch (group_norm -> silu -> conv1d, padding = dilation) to bf16 accuracy; the quad records against the stored output.  Dilations
    beyond the sample (every side tap padded), samples that are not multiples of the 128-row blocks (two samples in one workgroup),
    64- and 128-column workgroups."""
    import torch.nn.functional as F_
    g = torch.Generator().manual_seed(N * L + Cin + dil)
    x = (torch.randn(N * L, Cin, generator=g) * 1.5 + 0.3).to(BF).cuda()
    w = torch.randn(Cout, Cin, 3, generator=g) * (1.0 / (3 * Cin) ** 0.5)
    bias = torch.randn(Cout, generator=g).cuda()
    gamma, beta = (1 + 0.2 * torch.randn(Cin, generator=g)).cuda(), (0.2 * torch.randn(Cin, generator=g)).cuda()
    wp = ops.pack_conv_weight(w, BF).cuda()
    geom = ops.Geom.per_sample(N, L)
    a, b = ops.gn_stats(x, gamma, beta, geom)
    xn = ops.gn_apply(x, a, b, geom, act=True)
    want_stats = (N * L) % 64 == 0
    rec_ref = torch.zeros(N * L // 64, Cout // 4, 2, device="cuda") if want_stats else None
    y_ref = ops.conv_gemm(xn, wp, bias, taps=ops.taps_audio(dil), dims=(L, 1, 1), stats=rec_ref)
    rec = torch.full_like(rec_ref, 7.0) if want_stats else None
    y = torch.full((N * L, Cout), 3.0, dtype=BF, device="cuda")
    ops.aconv(x, a, b, wp, bias, N, L, dil, act=True, out=y, stats=rec)
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref), f"rel-L2 {rel_l2(y.float().cpu(), y_ref.float().cpu().numpy()):.3e}"
    # fp32 torch restatement of the reference operators on the same bf16-rounded input and weights
    xt = x.float().cpu().reshape(N, L, Cin).permute(0, 2, 1)
    ref = F_.conv1d(F_.silu(F_.group_norm(xt, 32, gamma.cpu(), beta.cpu(), eps=1e-5)), w.to(BF).float(), bias.cpu(), padding=dil, dilation=dil)
    ref = ref.permute(0, 2, 1).reshape(N * L, Cout)
    assert rel_l2(y.float().cpu(), ref.numpy()) < 1e-2
    if want_stats:
        yq = y.float().reshape(N * L // 64, 64, Cout // 4, 4)
        s1, s2 = yq.sum(dim=(1, 3)), (yq * yq).sum(dim=(1, 3))
        assert torch.allclose(rec[..., 0], s1, rtol=1e-4, atol=1e-2) and torch.allclose(rec[..., 1], s2, rtol=1e-4, atol=1e-2)
        # and the records are usable where the GEMM's are: same sums up to the order of the fp32 additions
        assert torch.allclose(rec, rec_ref, rtol=1e-4, atol=1e-2)


def test_aconv_column_slice_output_and_rejects(ops):
    """Output into the right-hand column slice of a wider buffer (the engine's concat views), input a column slice too; overlapping
    X / Y and unsupported shapes are refused by the C side."""
    from mm_diffusion import _hip as H
    N, L, Cin, Cout, dil = 2, 400, 128, 128, 4
    g = torch.Generator().manual_seed(11)
    wide_x = torch.randn(N * L, 2 * Cin, generator=g).to(BF).cuda()
    x = wide_x[:, Cin:]
    w = torch.randn(Cout, Cin, 3, generator=g) * 0.05
    wp = ops.pack_conv_weight(w, BF).cuda()
    gamma, beta = torch.ones(Cin).cuda(), torch.zeros(Cin).cuda()
    geom = ops.Geom.per_sample(N, L)
    a, b = ops.gn_stats(x, gamma, beta, geom)
    y_ref = ops.conv_gemm(ops.gn_apply(x, a, b, geom, act=True), wp, None, taps=ops.taps_audio(dil), dims=(L, 1, 1))
    wide_y = torch.zeros(N * L, 3 * Cout, dtype=BF, device="cuda")
    ops.aconv(x, a, b, wp, None, N, L, dil, out=wide_y[:, Cout:2 * Cout])
    torch.cuda.synchronize()
    assert torch.equal(wide_y[:, Cout:2 * Cout], y_ref) and float(wide_y[:, :Cout].abs().max()) == 0 and float(wide_y[:, 2 * Cout:].abs().max()) == 0
    with pytest.raises(H.MMDError):
        ops.aconv(x, a, b, wp, None, N, L, dil, out=wide_x[:, :Cin])      # overlapping ranges of one buffer
    with pytest.raises(H.MMDError):
        ops.aconv(x[:N * 100], a, b, wp, None, N, 100, dil)                # samples shorter than a row block


# --------------------------------------------------------------------------- row-strip GEMM, K = 128, one row fragment per wave (128-row blocks)
_STRIP_RF1_SCRIPT = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from mm_diffusion import ops
N, L, C = 2, 16384, 128
g = torch.Generator().manual_seed(5)
x = (torch.randn(N * L, C, generator=g) * 1.3 + 0.2).to(torch.bfloat16).cuda()
r = torch.randn(N * L, C, generator=g).to(torch.bfloat16).cuda()
w = (torch.randn(C, C, generator=g) * 0.09).to(torch.bfloat16).cuda()
bias = torch.randn(C, generator=g).cuda()
gamma, beta = (1 + 0.2 * torch.randn(C, generator=g)).cuda(), (0.2 * torch.randn(C, generator=g)).cuda()
geom = ops.Geom.per_sample(N, L)
a, b = ops.gn_stats(x, gamma, beta, geom)
rec = torch.zeros(N * L // 64, C // 4, 2, device="cuda")
y = ops.gn_conv1x1(x, a, b, geom, True, w, bias, residual=r, tile=131, stats=rec)
torch.cuda.synchronize()
torch.save({"y": y.cpu(), "rec": rec.cpu()}, sys.argv[2])
"""


def test_strip_k128_one_fragment_instance_matches_the_two_fragment_one(tmp_path):
    """conv1x1_strip_kernel<2, 1, 32> (MMD_STRIP_K128_RF1=1: 128-row blocks for the ds1 ResBlock out conv - norm + SiLU + 1x1 conv + skip +
    statistics, unet:457-476) against <2, 2, 64>: Y bitwise equal (same K order, same epilogue arithmetic), the quad records equal up to the
    order of the fp32 additions (a 64-row record is folded by a wave pair instead of one wave).  Each arm in its own interpreter: the
    switch is read once per process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "arm.py"
    script.write_text(_STRIP_RF1_SCRIPT)
    outs = {}
    for arm in ("0", "1"):
        env = dict(os.environ, MMD_STRIP_K128_RF1=arm)
        path = str(tmp_path / f"arm{arm}.pt")
        subprocess.run([sys.executable, str(script), os.path.join(root, "mm-diffusion_amd"), path], check=True, env=env, timeout=600)
        outs[arm] = torch.load(path)
    assert torch.equal(outs["0"]["y"], outs["1"]["y"])
    assert torch.allclose(outs["0"]["rec"], outs["1"]["rec"], rtol=1e-5, atol=1e-3)
    yq = outs["1"]["y"].float().reshape(-1, 64, 32, 4)
    assert torch.allclose(outs["1"]["rec"][..., 0], yq.sum(dim=(1, 3)), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("N,L,Cin,Cout,taps_kind", [
    (16, 6400, 256, 384, "1x1"),          # 102400 rows, 6 tiles: XCD-aware order (a multiple of 8 row splits), several co / ci tiles
    (8, 25600, 128, 128, "audio4"),       # 204800 rows, 3 taps of one tile: shifted rows with zero padding at every sample's ends
    (3, 1600, 384, 256, "audio4"),        # 4800 rows: fewer than 8 splits - the plain grid order
    (5, 12800, 128, 256, "1x1"),          # 64000 rows (not a multiple of the chunk grid), ragged last split
])
def test_wgrad_block_orders_match_torch(N, L, Cin, Cout, taps_kind):
    """mmd_conv_wgrad (wgrad_tr_bf16_kernel) in its XCD-aware block order (round 6: the blocks of a row split on one XCD, split counts from a
    cost model) and in the plain order it falls back to, against an fp64 torch reference of dW = dY^T gather(X) and db = colsum(dY) - every
    (co tile, ci tile, tap, split) must be covered exactly once whatever the order."""
    from mm_diffusion import ops
    M = N * L
    g = torch.Generator(device="cuda").manual_seed(M + Cin)
    x = torch.randn(M, Cin, device="cuda", generator=g).to(torch.bfloat16)
    dy = torch.randn(M, Cout, device="cuda", generator=g).to(torch.bfloat16)
    taps = ops.TAPS_1 if taps_kind == "1x1" else ops.taps_audio(4)
    dims = (1, 1, 1) if taps_kind == "1x1" else (L, 1, 1)
    dW = torch.zeros(Cout, Cin * len(taps), device="cuda")
    db = torch.zeros(Cout, device="cuda")
    ops.conv_wgrad(dy, x, dW, db, taps, dims)
    xd, dyd = x.double(), dy.double()
    ref = []
    for (o0, _, _) in taps:
        xs = torch.zeros_like(xd)
        pos = torch.arange(M, device="cuda") % L if taps_kind != "1x1" else None
        if o0 == 0:
            xs = xd
        else:
            ok = (pos + o0 >= 0) & (pos + o0 < L)
            src = (torch.arange(M, device="cuda") + o0).clamp(0, M - 1)
            xs = torch.where(ok[:, None], xd[src], torch.zeros_like(xd))
        ref.append(dyd.t() @ xs)
    ref = torch.cat(ref, dim=1)
    assert float((dW.double() - ref).norm() / ref.norm()) < 2e-6          # fp32 accumulation of exact bf16 products, atomics in any order
    assert float((db.double() - dyd.sum(0)).norm() / dyd.sum(0).norm()) < 2e-6


@pytest.mark.parametrize("seed", range(24))
def test_strip_pipeline_random_shapes(seed):
    """The software-pipelined row-strip instances (K = 256 without a residual, K = 384 / 512 with and without) on random shapes: odd and even
    numbers of 32-column sub-tiles, one sub-tile per block (deep column splits of few rows), ragged last row blocks, residual, fused
    GroupNorm (+ SiLU), statistics records - Y bitwise against the tiled kernels (tile 129; gn_apply + tile 129 for the fused norm), the
    records against an fp64 reduction of the stored values."""
    import random
    from mm_diffusion import ops
    rnd = random.Random(1000 + seed)
    K = rnd.choice([256, 384, 512, 384, 512])
    cc = 64 if K == 256 else 32
    Cout = cc * rnd.randint(1, 20 if K == 256 else 40)
    gn = rnd.random() < 0.5
    stats = rnd.random() < 0.5
    res = rnd.random() < 0.6
    if gn:
        S, Tn = rnd.randint(1, 5), 256 * rnd.randint(1, 12)
        M = S * Tn
    else:
        M = 64 * rnd.randint(1, 300) if stats else rnd.randint(33, 20000)
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = (torch.randn(M, K, device="cuda", generator=g) * 1.2 + 0.1).to(torch.bfloat16)
    w = (torch.randn(Cout, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(Cout, device="cuda", generator=g)
    r = torch.randn(M, Cout, device="cuda", generator=g).to(torch.bfloat16) if res else None
    rec = torch.full((M // 64, Cout // 4, 2), 3.0, device="cuda") if stats else None
    y = torch.full((M, Cout), float("nan"), device="cuda", dtype=torch.bfloat16)
    if gn:
        act = rnd.random() < 0.5
        gamma, beta = 1 + 0.1 * torch.randn(K, device="cuda", generator=g), torch.randn(K, device="cuda", generator=g)
        geom = ops.Geom.per_sample(S, Tn)
        ga, gb = ops.gn_stats(x, gamma, beta, geom)
        ops.gn_conv1x1(x, ga, gb, geom, act, w, b, residual=r, tile=131, out=y, stats=rec)
        ref = ops.conv_gemm(ops.gn_apply(x, ga, gb, geom, act=act), w, b, residual=r, tile=129)
    else:
        ops.conv_gemm(x, w, b, residual=r, tile=131, out=y, stats=rec)
        ref = ops.conv_gemm(x, w, b, residual=r, tile=129)
    assert torch.equal(y.view(torch.int16), ref.view(torch.int16)), (K, Cout, M, gn, stats, res)
    if stats:
        yf = y.double().view(M // 64, 64, Cout // 4, 4)
        want = torch.stack([yf.sum((1, 3)), (yf * yf).sum((1, 3))], dim=-1)
        assert float((rec.double() - want).abs().max() / want.abs().max()) < 2e-6, (K, Cout, M, gn, res)


@pytest.mark.parametrize("gn", [None, "silu", "affine"])
@pytest.mark.parametrize("N,H,W,Cin", [(1, 8, 8, 32), (2, 64, 72, 96), (1, 16, 16, 128), (3, 64, 64, 64), (1, 64, 64, 256), (4, 64, 64, 128), (2, 64, 64, 384)])
def test_vconv_two_slot_ring_equals_three_slot_ring(gn, N, H, W, Cin):
    """The fused VideoConv's two-slot weight ring (MMD_VCONV_RING=2, 121.75 KB of LDS, weights one step ahead, input norm re-timed) against the
    three-slot ring (the default): output and statistics records bitwise equal - odd and even numbers of 32-channel chunks (the ring's
    slot parity flips per patch when the step count is odd), one to nine patches per persistent block."""
    import os
    from mm_diffusion import ops
    from test_vconv_gpu import make
    x, ws, wt, bs, bt, a, b = make(N, H, W, Cin, seed=N * 1000 + Cin)
    wf = ops.vconv_pack(ops.pack_conv_weight(ws.float(), torch.bfloat16), ops.pack_conv_weight(wt.float(), torch.bfloat16))
    geom = ops.Geom.per_sample(N, 16 * H * W)
    kw = {} if gn is None else dict(a=a, b=b, geom=geom, act=gn == "silu")
    M = N * 16 * H * W
    out = {}
    old = os.environ.get("MMD_VCONV_RING")
    try:
        for ring in ("3", "2"):
            os.environ["MMD_VCONV_RING"] = ring
            rec = torch.full((M // 64, 32, 2), float("nan"), device="cuda")
            y = ops.vconv2d1d(x, wf, bs, bt, N, 16, H, W, stats=rec, **kw)
            torch.cuda.synchronize()
            out[ring] = (y, rec)
    finally:
        if old is None:
            os.environ.pop("MMD_VCONV_RING", None)
        else:
            os.environ["MMD_VCONV_RING"] = old
    assert torch.isfinite(out["2"][0].float()).all()
    assert torch.equal(out["2"][0].view(torch.int16), out["3"][0].view(torch.int16))
    assert torch.equal(out["2"][1], out["3"][1])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("S,Tn,C,pad,film,act", [(4, 400, 512, 0, False, True), (2, 400, 1024, 0, True, True), (2, 400, 896, 128, True, False),
                                                  (3, 1024, 512, 0, False, True), (2, 100, 2048, 0, False, False), (5, 37, 128, 0, True, True),
                                                  (1, 1600, 256, 64, False, True)])
def test_gn_group_one_launch(dtype, S, Tn, C, pad, film, act):
    """mmd_gn_group (GroupNorm32 (+FiLM) (+SiLU) of a few hundred rows per slice in one launch) against an fp64 torch GroupNorm of the
    same slices, its affine against mmd_gn_stats', and its apply step bitwise against mmd_gn_apply fed with its own affine."""
    from mm_diffusion import ops
    g = torch.Generator(device="cuda").manual_seed(S * 1000 + Tn + C)
    M = S * Tn
    buf = (torch.randn(M, C + pad, device="cuda", generator=g) * 1.5 + 0.7).to(dtype)
    x = buf[:, :C]
    gamma = 1 + 0.2 * torch.randn(C, device="cuda", generator=g)
    beta = 0.3 * torch.randn(C, device="cuda", generator=g)
    fl = 0.3 * torch.randn(S, 2 * C, device="cuda", generator=g) if film else None
    geom = ops.Geom.per_sample(S, Tn)
    assert ops.gn_group_ok(x, geom)
    a = torch.full((S, C), float("nan"), device="cuda")
    b = torch.full((S, C), float("nan"), device="cuda")
    y = torch.full((M, C), float("nan"), device="cuda", dtype=dtype)
    ops.gn_group(x, gamma, beta, geom, film=fl, a=a, b=b, out=y, act=act)
    a2, b2 = ops.gn_group(x, gamma, beta, geom, film=fl)                      # affine only
    y3 = ops.gn_group(x, gamma, beta, geom, film=fl, out=torch.empty_like(y), act=act)   # tensor only
    assert torch.equal(a, a2) and torch.equal(b, b2) and torch.equal(y, y3)
    xd = x.double().view(S, Tn, 32, C // 32)
    mean = xd.mean(dim=(1, 3), keepdim=True)
    var = xd.var(dim=(1, 3), unbiased=False, keepdim=True)
    ref = ((xd - mean) / torch.sqrt(var + 1e-5)).view(S, Tn, C) * gamma.double() + beta.double()
    if film:
        ref = ref * (1 + fl[:, None, :C].double()) + fl[:, None, C:].double()
    if act:
        ref = ref * torch.sigmoid(ref)
    ref = ref.view(M, C)
    err = float((y.double() - ref).norm() / ref.norm())
    assert err < (4e-3 if dtype == torch.bfloat16 else 2e-6), err
    a0, b0 = ops.gn_stats(x, gamma, beta, geom, film=fl)
    assert float((a - a0).abs().max() / a0.abs().max()) < 2e-6 and float((b - b0).abs().max() / b0.abs().max()) < 1e-5
    y1 = ops.gn_apply(x, a, b, geom, act=act)
    assert torch.equal(y.view(torch.int16 if dtype == torch.bfloat16 else torch.int32), y1.view(torch.int16 if dtype == torch.bfloat16 else torch.int32))


def test_denoised_fn_in_p_mean_variance_and_p_sample():
    """denoised_fn (reference gd:263-268: applied to the x_0 prediction before the clamp) through the fused update kernel - the unclamped
    prediction, the caller's function, then the kernel again reading the processed tensor as an x_0 prediction - against the reference
    fixture pmv_denoised.npz (p_mean_variance with clip on / off, p_sample with the fixture's noise)."""
    import numpy as np
    from helpers import gold, rel_l2
    from mm_diffusion import multimodal_script_util as msu
    g = gold("pmv_denoised")
    f = msu.model_and_diffusion_defaults()
    f.update(timestep_respacing="10", learn_sigma=True)
    diff = msu.create_gaussian_diffusion(steps=f["diffusion_steps"], learn_sigma=True, noise_schedule=f["noise_schedule"],
                                         timestep_respacing="10")
    dev = torch.device("cuda")
    x = {"video": torch.from_numpy(g["xv"]).to(dev), "audio": torch.from_numpy(g["xa"]).to(dev)}
    vo, ao = torch.from_numpy(g["vo"]).to(dev), torch.from_numpy(g["ao"]).to(dev)
    t = torch.from_numpy(g["t"]).to(dev)
    model = lambda v, a, ts, **kw: (vo, ao)      # noqa: E731
    fn = lambda z: 0.5 * z + 0.1                  # noqa: E731
    for clip in (1, 0):
        out = diff.p_mean_variance(model, x, t, clip_denoised=bool(clip), denoised_fn=fn)
        for k in ("mean", "log_variance", "pred_xstart"):
            for key in ("video", "audio"):
                assert rel_l2(out[k][key].cpu(), g[f"{k}_{key}_clip{clip}"]) < 1e-6, (k, key, clip)
    noise = iter([torch.from_numpy(g["noise_v"]).to(dev), torch.from_numpy(g["noise_a"]).to(dev)])
    diff.noise_source = lambda like: next(noise)
    ps = diff.p_sample(model, x, t, clip_denoised=True, denoised_fn=fn)
    assert rel_l2(ps["sample"]["video"].cpu(), g["sample_video"]) < 1e-6 and rel_l2(ps["sample"]["audio"].cpu(), g["sample_audio"]) < 1e-6


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_weight_pack_and_grad_unpack_tiles(dtype):
    """mmd_pack_conv_weights / mmd_unpack_conv_grads on LDS tiles (32 output x (216 / taps) input channels x all taps): every conv weight of a
    mixed list - edge tiles, 1 / 3 / 9 / 27 taps, 1 and 3 input / output channels like the stems and heads - against torch permutes, the
    packed gradient accumulators folded into .grad (+=) and cleared."""
    from mm_diffusion.optim import WeightPacker
    g = torch.Generator(device="cuda").manual_seed(5)
    shapes = [(128, 3, 3, 3, 3), (3, 128, 3, 3, 3), (128, 1, 3), (2, 128, 3), (256, 384, 3, 3), (40, 24, 3), (512, 512, 1), (33, 17, 1, 3, 3), (384, 640, 3)]
    params = []
    for sh in shapes:
        p = torch.randn(*sh, device="cuda", generator=g).requires_grad_(True)
        p.grad = torch.randn(*sh, device="cuda", generator=g)
        params.append(p)
    bias = torch.randn(7, device="cuda").requires_grad_(True)          # not a conv weight: ignored by the packer
    bias.grad = torch.zeros(7, device="cuda")
    wp = WeightPacker(params + [bias], dtype)
    for p in params:
        Cout, Cin = p.shape[0], p.shape[1]
        nt = p.numel() // (Cout * Cin)
        w = p.detach().reshape(Cout, Cin, nt)
        fwd, bwd = p._mmd_packed
        assert torch.equal(fwd, w.permute(0, 2, 1).reshape(Cout, nt * Cin).to(dtype)), tuple(p.shape)
        assert torch.equal(bwd, w.permute(1, 2, 0).reshape(Cin, nt * Cout).to(dtype)), tuple(p.shape)
    want = []
    for p in params:
        Cout, Cin = p.shape[0], p.shape[1]
        nt = p.numel() // (Cout * Cin)
        acc = torch.randn(Cout, nt * Cin, device="cuda", generator=g)
        p._mmd_wgrad.copy_(acc)
        want.append(p.grad.clone() + acc.view(Cout, nt, Cin).permute(0, 2, 1).reshape(p.shape))
    wp.fold_grads()
    for p, w in zip(params, want):
        assert torch.equal(p.grad, w), tuple(p.shape)
        assert not p._mmd_wgrad.any()
    # a weight changed in place is re-packed by refresh()
    with torch.no_grad():
        params[4].mul_(0.5)
    wp.refresh()
    w = params[4].detach().reshape(256, 384, 9)
    assert torch.equal(params[4]._mmd_packed[0], w.permute(0, 2, 1).reshape(256, 9 * 384).to(dtype))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gn_bwd_with_a_kept_zero_workspace(dtype, monkeypatch):
    """mmd_gn_bwd_ws0 (the workspace stays zero between calls: no fill launch per norm) against mmd_gn_bwd on the same inputs, called
    three times in a row on ONE kept workspace (different inputs each time: a stale accumulator would show), and the workspace is zero
    again afterwards."""
    from mm_diffusion import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    S, Tn, C = 3, 640, 256
    geom = ops.Geom.per_sample(S, Tn)
    gamma, beta = 1 + 0.1 * torch.randn(C, device="cuda", generator=g), 0.2 * torch.randn(C, device="cuda", generator=g)
    film = 0.2 * torch.randn(S, 2 * C, device="cuda", generator=g)
    for it in range(3):
        x = (torch.randn(S * Tn, C, device="cuda", generator=g) * (1 + it) + 0.3).to(dtype)
        dy = torch.randn(S * Tn, C, device="cuda", generator=g).to(dtype)
        mr = torch.empty(S, 32, 2, device="cuda")
        a, b = ops.gn_stats(x, gamma, beta, geom, film=film, mr=mr)
        outs = []
        for ws0 in (False, True):
            monkeypatch.setattr(ops, "_GN_BWD_WS0", ws0)
            dx = torch.full_like(x, float("nan"))
            dgam, dbet, dfilm = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(S, 2 * C, device="cuda")
            ops.gn_bwd(x, dy, dx, geom, a, b, mr, gamma, beta, film, True, dgam, dbet, dfilm)
            outs.append((dx.float(), dgam, dbet, dfilm))
        for u, v in zip(outs[0], outs[1]):
            assert float((u - v).norm() / v.norm()) < (2e-3 if dtype == torch.bfloat16 else 1e-5), it      # (atomic accumulation order differs run to run)
    assert ops._gn_bwd_ws and all(not w[: w.numel() - 64 * 3].any() for k, w in ops._gn_bwd_ws.items() if k[2] == S * C * 2 + S * 64)
