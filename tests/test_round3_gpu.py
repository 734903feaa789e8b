"""Round-3 kernels through the C ABI.

* mmd_gn_conv_gemm: GroupNorm(+FiLM)(+SiLU) of a 3x3 conv's input applied to the staged halo in LDS (tile 130).  Pinned two ways:
  bitwise against the two launches it replaces (mmd_gn_apply + mmd_conv_gemm tile 130: same expressions, same rounding points - both
  are themselves checked against the oracle's GroupNorm / conv primitives in test_ops_gpu.py), and against the fp32 torch
  restatement of norm -> SiLU -> conv3d on the same bf16-rounded inputs (rel-L2 <= 1e-2, the bf16 bound of test_ops_gpu.py).
"""
import pytest
import torch
import torch.nn.functional as F_

from helpers import rel_l2

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    from mm_diffusion import ops as o
    return o


@pytest.mark.parametrize("act", [True, False])
@pytest.mark.parametrize("N,F,H,W,Cin,Cout,cat", [(2, 3, 32, 32, 64, 128, 0), (1, 2, 16, 64, 128, 96, 0), (2, 2, 32, 32, 192, 264, 64),
                                                 (1, 2, 64, 64, 384, 128, 0)])
def test_halo_fused_groupnorm(ops, act, N, F, H, W, Cin, Cout, cat):
    """Several channel chunks (the affine ring and the in-loop transform of chunk c + 1 under the MFMAs of chunk c), patches on every
    border (padding pixels must stay zero AFTER the normalisation), a ragged Cout, the input as a column slice of a wider buffer."""
    M = N * F * H * W
    g = torch.Generator(device="cuda").manual_seed(M + Cin)
    xb = (torch.randn(M, Cin + cat, device="cuda", generator=g) * 1.5 + 0.3).to(BF)
    x = xb[:, cat:]
    w = (torch.randn(Cout, Cin * 9, device="cuda", generator=g) * (Cin * 9) ** -0.5).to(BF)
    b = torch.randn(Cout, device="cuda", generator=g)
    gamma = 1 + 0.2 * torch.randn(Cin, device="cuda", generator=g)
    beta = 0.2 * torch.randn(Cin, device="cuda", generator=g)
    geom = ops.Geom.per_sample(N, F * H * W)
    ga, gb = ops.gn_stats(x, gamma, beta, geom)
    dims = (N * F, H, W)
    assert ops.halo_tile_ok(x, ops.TAPS_SPATIAL, dims)
    y = ops.gn_conv_gemm(x, ga, gb, geom, act, w, b, ops.TAPS_SPATIAL, dims, tile=130)
    if H % 16 == 0 and W % 16 == 0:        # the 16 x 16-patch halo tile: same K order, same transform -> bitwise the same
        y16 = ops.gn_conv_gemm(x, ga, gb, geom, act, w, b, ops.TAPS_SPATIAL, dims, tile=133)
        assert torch.equal(y.view(torch.int16), y16.view(torch.int16)), f"tile 133 vs 130 (fused norm): rel-L2 {rel_l2(y16.float().cpu(), y.float().cpu().numpy()):.3e}"
    xn = ops.gn_apply(x, ga, gb, geom, act=act)
    y2 = ops.conv_gemm(xn, w, b, taps=ops.TAPS_SPATIAL, dims=dims, tile=130)
    torch.cuda.synchronize()
    e_pair = rel_l2(y.float().cpu(), y2.float().cpu().numpy())
    assert torch.equal(y.view(torch.int16), y2.view(torch.int16)), f"fused halo GroupNorm differs from gn_apply + tile 130: rel-L2 {e_pair:.3e}"
    # fp32 restatement on the same rounded inputs
    xf = x.float().cpu().reshape(N, F, H, W, Cin).permute(0, 4, 1, 2, 3)                       # [N, C, F, H, W]
    ref = F_.group_norm(xf, 32, gamma.cpu(), beta.cpu(), eps=1e-5)
    ref = F_.silu(ref) if act else ref
    wt = w.float().cpu().reshape(Cout, 3, 3, Cin).permute(0, 3, 1, 2)                          # packed K = tap * Cin + ci
    ref = F_.conv3d(ref, wt[:, :, None], b.cpu(), padding=(0, 1, 1))
    got = y.float().cpu().reshape(N, F, H, W, Cout).permute(0, 4, 1, 2, 3)
    assert rel_l2(got, ref) < 1e-2


def test_engine_uses_the_fused_halo_norm_and_matches_the_unfused_plan(monkeypatch):
    """The launch plan of the mid-size model with and without MMD_HALO_GN: the fused plan really carries gn_conv_gemm launches, and the
    outputs agree to bf16 rounding noise.  (The op itself is bitwise the two-launch form - test_halo_gn_* above; whole plans are not,
    because the unfused plan is free to autotune another tile for the producer of a norm's records, and tiles sum a record's 64 rows
    in different orders.)"""
    import importlib
    from helpers import flags, inputs
    from mm_diffusion import multimodal_script_util as msu, ops as o
    from mm_diffusion.synth import synth_init_
    fl = flags("mid", use_fp16=True)
    outs = []
    for on in (True, False):
        monkeypatch.setattr(o, "_HALO_GN", on)
        model, _ = msu.create_model_and_diffusion(**fl)
        synth_init_(model)
        model.cuda().eval()
        v, a = inputs(fl, 2, 3)
        import random
        random.seed(5)
        with torch.no_grad():
            ov, oa = model(v.cuda(), a.cuda(), torch.tensor([17, 400]).cuda())
        eng = next(iter(model._engines.values()))
        names = [e[2] for e in eng.plan]
        outs.append((ov.clone(), oa.clone(), names.count("mmd_gn_conv_gemm"), names.count("mmd_gn_apply")))
        model.release_engines()
    assert outs[0][2] > 0 and outs[1][2] == 0 and outs[0][3] < outs[1][3]
    assert rel_l2(outs[0][0].cpu(), outs[1][0].cpu().numpy()) < 5e-3 and rel_l2(outs[0][1].cpu(), outs[1][1].cpu().numpy()) < 5e-3


@pytest.mark.parametrize("dt", [torch.float32, BF])
@pytest.mark.parametrize("res", [False, True])
@pytest.mark.parametrize("M,Cin,Cout,taps", [(4096, 512, 512, "3x3"), (128 * 9 + 37, 64, 136, "1"), (1600, 128, 256, "a2"), (700, 256, 64, "1"),
                                            (2 * 6 * 16 * 16, 192, 384, "3x3"), (300, 64, 64, "a400")])
def test_deep_ring_tile_is_bitwise_the_tiled_loop(ops, dt, res, M, Cin, Cout, taps):
    """conv_gemm tile 132 (four-slot LDS ring, counted vmcnt waits, one raw barrier per K step) against the register-staged 128 tile:
    same K order and epilogue, so bitwise equal - with 1, 2, 3 (fewer than the ring depth) and many K steps, ragged M / Cout edges,
    padding taps (3x3 borders, a dilation beyond the sequence), a residual, fp32 (32-channel K steps) and bf16."""
    if taps == "3x3":
        side = 8 if M == 4096 else 16
        tp, dims = ops.TAPS_SPATIAL, (M // (side * side), side, side)
    elif taps == "1":
        tp, dims = ops.TAPS_1, (1, 1, 1)
    else:
        tp, dims = ops.taps_audio(int(taps[1:])), (M // 2 if taps == "a2" else M, 1, 1)
    g = torch.Generator(device="cuda").manual_seed(M + Cin)
    x = torch.randn(M, Cin, device="cuda", generator=g).to(dt)
    w = (torch.randn(Cout, Cin * len(tp), device="cuda", generator=g) * (Cin * len(tp)) ** -0.5).to(dt)
    b = torch.randn(Cout, device="cuda", generator=g)
    r = torch.randn(M, Cout, device="cuda", generator=g).to(dt) if res else None
    y0 = ops.conv_gemm(x, w, b, taps=tp, dims=dims, residual=r, tile=128)
    for _ in range(3):
        y1 = ops.conv_gemm(x, w, b, taps=tp, dims=dims, residual=r, tile=132)
        assert torch.equal(y0, y1), f"rel-L2 {rel_l2(y1.float().cpu(), y0.float().cpu().numpy()):.3e}"


def test_deep_ring_tile_statistics(ops):
    """The ring tile shares the 128-row epilogue: output statistics records bitwise equal to tile 129's."""
    M, Cin, Cout = 4096, 512, 512
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(M, Cin, device="cuda", generator=g).to(BF)
    w = (torch.randn(Cout, Cin * 3, device="cuda", generator=g) * (Cin * 3) ** -0.5).to(BF)
    b = torch.randn(Cout, device="cuda", generator=g)
    recs = []
    for tile in (129, 132):
        rec = torch.zeros(M // 64, Cout // 4, 2, device="cuda")
        y = ops.conv_gemm(x, w, b, taps=ops.TAPS_TEMPORAL, dims=(16, 64, 1), tile=tile, stats=rec)
        recs.append((y.clone(), rec))
    assert torch.equal(recs[0][0], recs[1][0]) and torch.equal(recs[0][1], recs[1][1])
    ysum = recs[0][0].float().reshape(M // 64, 64, Cout // 4, 4).sum((1, 3))
    assert rel_l2(recs[1][1][:, :, 0].cpu(), ysum.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("res", [False, True])
@pytest.mark.parametrize("D0,H,W,Cin,Cout", [(5, 16, 32, 64, 128), (3, 16, 16, 128, 96), (2, 32, 48, 192, 256 + 8), (1, 64, 64, 640, 128)])
def test_halo16_tile_is_bitwise_the_halo_tile(ops, res, D0, H, W, Cin, Cout):
    """conv_gemm tile 133 (16 x 16 patches, 8 waves, three-slot weight ring, counted waits) against tile 130: chunk-major K order in
    both, so bitwise equal - 1 to 10 channel chunks, patches on every border, a single-patch frame, ragged Cout, a residual; and
    against the tap-major direct-to-LDS loop to rounding."""
    M = D0 * H * W
    g = torch.Generator(device="cuda").manual_seed(M + Cin)
    x = torch.randn(M, Cin, device="cuda", generator=g).to(BF)
    w = (torch.randn(Cout, Cin * 9, device="cuda", generator=g) * (Cin * 9) ** -0.5).to(BF)
    b = torch.randn(Cout, device="cuda", generator=g)
    r = torch.randn(M, Cout, device="cuda", generator=g).to(BF) if res else None
    y0 = ops.conv_gemm(x, w, b, taps=ops.TAPS_SPATIAL, dims=(D0, H, W), residual=r, tile=130)
    for _ in range(3):
        y1 = ops.conv_gemm(x, w, b, taps=ops.TAPS_SPATIAL, dims=(D0, H, W), residual=r, tile=133)
        assert torch.equal(y0, y1), f"rel-L2 {rel_l2(y1.float().cpu(), y0.float().cpu().numpy()):.3e}"
    y2 = ops.conv_gemm(x, w, b, taps=ops.TAPS_SPATIAL, dims=(D0, H, W), residual=r, tile=129)
    assert rel_l2(y1.float().cpu(), y2.float().cpu().numpy()) < 4e-3


# ---------------------------------------------------------------------------------------------------------------- in-launch GroupNorm tails
def _tail(ops, S, rows, Cbuf, c0, C=None, fq0=0, gamma=None, beta=None, film=None, finalize=True):
    """A GnTail over fresh accumulators / counters / affine outputs (kept alive on the struct)."""
    from mm_diffusion import _hip as H
    st = H.GnTail()
    st._acc = torch.zeros(S * (Cbuf // 4) * 4, dtype=torch.int64, device="cuda")
    st._cnt = torch.zeros(4, dtype=torch.int32, device="cuda")
    st.acc, st.q_ld, st.q_off, st.S, st.rows_per_slice = st._acc.data_ptr(), Cbuf // 4, c0 // 4, S, rows
    if finalize:
        st._a, st._b = torch.zeros(S, C, device="cuda"), torch.zeros(S, C, device="cuda")
        st.launch_counter, st.shared_counter, st.n_producers = st._cnt.data_ptr(), st._cnt.data_ptr() + 4, 1
        st.C, st.fq0, st.gamma, st.beta, st.eps = C, fq0, gamma.data_ptr(), beta.data_ptr(), ops.GN_EPS
        st.film, st.film_ld = (0 if film is None else film.data_ptr()), (0 if film is None else film.stride(0))
        st.a_out, st.b_out = st._a.data_ptr(), st._b.data_ptr()
    return st


@pytest.mark.parametrize("tile", [64, 128, 129, 131, 132])
@pytest.mark.parametrize("S,Tn,Cin,Cout,taps3,res,use_film", [(4, 256, 128, 128, False, True, True), (2, 1024, 256, 256, False, False, False),
                                                             (8, 64, 512, 512, False, True, False), (2, 512, 128, 384, True, False, True),
                                                             (3, 192, 384, 128, False, False, False)])
def test_tail_affine_matches_the_statistics_pass(ops, tile, S, Tn, Cin, Cout, taps3, res, use_film):
    """mmd_conv_gemm_tail on every row-tiled loop: Y bitwise the plain kernel's; the affine its LAST block leaves equals mmd_gn_stats
    over Y (rel 2e-5: integer-exact sums vs the pivoted fp32 pass) and is BITWISE the same from every tile family (the totals are
    integer sums of exactly-converted partials... of partials that differ per family, so only equal to rounding across families)."""
    M = S * Tn
    taps, dims = (ops.TAPS_TEMPORAL, (16, M // 16, 1)) if taps3 else (ops.TAPS_1, (1, 1, 1))
    if tile == 131 and not ops.strip_tile_ok(torch.empty(M, Cin, dtype=BF), Cout, taps, stats=True):
        pytest.skip("shape outside the strip kernel")
    g = torch.Generator(device="cuda").manual_seed(M + Cout)
    x = torch.randn(M, Cin, device="cuda", generator=g).to(BF)
    w = (torch.randn(Cout, Cin * len(taps), device="cuda", generator=g) * (Cin * len(taps)) ** -0.5).to(BF)
    b = torch.randn(Cout, device="cuda", generator=g)
    r = torch.randn(M, Cout, device="cuda", generator=g).to(BF) if res else None
    gamma, beta = 1 + 0.3 * torch.randn(Cout, device="cuda", generator=g), 0.3 * torch.randn(Cout, device="cuda", generator=g)
    film = 0.2 * torch.randn(S, 2 * Cout, device="cuda", generator=g) if use_film else None
    if Cout % 128:
        pytest.skip("consumer channels must be a multiple of 128")
    st = _tail(ops, S, Tn, Cout, 0, C=Cout, gamma=gamma, beta=beta, film=film)
    y0 = ops.conv_gemm(x, w, b, taps=taps, dims=dims, residual=r, tile=tile)
    y1 = ops.conv_gemm(x, w, b, taps=taps, dims=dims, residual=r, tile=tile, tail=st)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    geom = ops.Geom.per_sample(S, Tn)
    a_ref, b_ref = ops.gn_stats(y1, gamma, beta, geom, film=film)
    assert int(st._cnt[0]) > 0 and int(st._cnt[1]) == 1
    assert rel_l2(st._a.cpu(), a_ref.cpu().numpy()) < 2e-5 and rel_l2(st._b.cpu(), b_ref.cpu().numpy()) < 2e-5
    # exact totals: the accumulators hold the sums of the STORED values
    yf = y1.float().reshape(S, Tn, Cout // 4, 4)
    acc = st._acc.reshape(S, Cout // 4, 4).double().cpu()
    tot = acc[..., 0] / 2 ** 8 + acc[..., 1] / 2 ** 32
    tsq = acc[..., 2] / 2 ** 4 + acc[..., 3] / 2 ** 28
    assert rel_l2(tot.float(), yf.sum((1, 3)).cpu().numpy()) < 1e-5 and rel_l2(tsq.float(), (yf * yf).sum((1, 3)).cpu().numpy()) < 1e-5
    # repeatable to the last bit (atomics in any order: integer sums)
    st2 = _tail(ops, S, Tn, Cout, 0, C=Cout, gamma=gamma, beta=beta, film=film)
    ops.conv_gemm(x, w, b, taps=taps, dims=dims, residual=r, tile=tile, tail=st2)
    torch.cuda.synchronize()
    assert torch.equal(st._acc, st2._acc) and torch.equal(st._a, st2._a) and torch.equal(st._b, st2._b)


def test_tail_two_producers_of_a_concat_and_two_consumers(ops):
    """The skip-concat case: P writes the right half of a buffer (and finalises the norm over ITS columns, the next input block's),
    later Q writes the left half and finalises the norm over the WHOLE buffer from quads P accumulated long before; a GroupNorm-fused
    strip launch as a producer; batch invariance: the rows of sample 0 give the same affine when run alone."""
    S, Tn, CL, CR = 2, 512, 256, 128
    M, Cb = S * Tn, CL + CR
    g = torch.Generator(device="cuda").manual_seed(5)
    buf = torch.zeros(M, Cb, device="cuda", dtype=BF)
    xP, xQ = torch.randn(M, 128, device="cuda", generator=g).to(BF), torch.randn(M, 256, device="cuda", generator=g).to(BF)
    wP = (torch.randn(CR, 128, device="cuda", generator=g) * 128 ** -0.5).to(BF)
    wQ = (torch.randn(CL, 256, device="cuda", generator=g) * 256 ** -0.5).to(BF)
    bP, bQ = torch.randn(CR, device="cuda", generator=g), torch.randn(CL, device="cuda", generator=g)
    gR, bR = 1 + 0.1 * torch.randn(CR, device="cuda", generator=g), 0.1 * torch.randn(CR, device="cuda", generator=g)
    gA, bA = 1 + 0.1 * torch.randn(Cb, device="cuda", generator=g), 0.1 * torch.randn(Cb, device="cuda", generator=g)
    geom = ops.Geom.per_sample(S, Tn)
    gna, gnb = ops.gn_stats(xP, torch.ones(128, device="cuda"), torch.zeros(128, device="cuda"), geom)

    def run(S_, rows):
        stP = _tail(ops, S_, Tn, Cb, CL, C=CR, fq0=CL // 4, gamma=gR, beta=bR)
        stQ = _tail(ops, S_, Tn, Cb, 0, C=Cb, fq0=0, gamma=gA, beta=bA)
        stQ.acc, stQ._acc = stP.acc, stP._acc                     # one accumulator array per buffer
        bb = buf[:rows]
        ge = ops.Geom.per_sample(S_, Tn)
        ops.gn_conv1x1(xP[:rows], gna[:S_], gnb[:S_], ge, True, wP, bP, out=bb[:, CL:], tail=stP)       # strip with fused norm
        ops.conv_gemm(xQ[:rows], wQ, bQ, out=bb[:, :CL], tail=stQ, tile=129)
        torch.cuda.synchronize()
        return stP, stQ, bb

    stP, stQ, bb = run(S, M)
    aR, bR_ = ops.gn_stats(bb[:, CL:], gR, bR, geom)
    aA, bA_ = ops.gn_stats(bb, gA, bA, geom)
    assert rel_l2(stP._a.cpu(), aR.cpu().numpy()) < 2e-5 and rel_l2(stP._b.cpu(), bR_.cpu().numpy()) < 2e-5
    assert rel_l2(stQ._a.cpu(), aA.cpu().numpy()) < 2e-5 and rel_l2(stQ._b.cpu(), bA_.cpu().numpy()) < 2e-5
    sP1, sQ1, _ = run(1, Tn)
    assert torch.equal(sP1._a[0], stP._a[0]) and torch.equal(sQ1._a[0], stQ._a[0]) and torch.equal(sQ1._b[0], stQ._b[0])


# (test_engine_tails_match_the_record_path - the ENGINE mode that used the tails, MMD_GN_TAIL - went with that mode in round 5: measured slower in
# round 3, its premise measured false in round 5 (profiles/r05_chain_interference_and_launch_modes.txt), and its batch-row check was the one test
# of the suite that ever failed without a reproduction: once in 23 runs, inside a 380-test process.  The entry points above stay, with their tests.)


@pytest.mark.parametrize("dt", [torch.float32, BF])
@pytest.mark.parametrize("act", [False, True])
@pytest.mark.parametrize("N,F,HW,C", [(2, 16, 64, 256), (1, 16, 37, 384), (2, 8, 16, 128), (1, 16, 20, 512), (1, 5, 9, 1024)])
def test_gn_small_is_stats_plus_apply(ops, dt, act, N, F, HW, C):
    """mmd_gn_small (one pass, register-resident two-pass statistics) on temporal slices (the frames of a pixel, strided rows) against
    torch's group_norm on the same values, and against gn_stats + gn_apply; ragged slice counts, every group size (4 ... 32 channels)."""
    g = torch.Generator(device="cuda").manual_seed(C + HW)
    x = (torch.randn(N * F * HW, C, device="cuda", generator=g) * 2 + 0.7).to(dt)
    gamma, beta = 1 + 0.3 * torch.randn(C, device="cuda", generator=g), 0.3 * torch.randn(C, device="cuda", generator=g)
    geom = ops.Geom.temporal(N, F, HW)
    assert ops.gn_small_ok(x, geom)
    y = ops.gn_small(x, gamma, beta, geom, act=act)
    a, b = ops.gn_stats(x, gamma, beta, geom)
    y2 = ops.gn_apply(x, a, b, geom, act=act)
    xt = x.float().reshape(N, F, HW, C).permute(0, 2, 3, 1).reshape(N * HW, C, F)         # slices (n, pixel): [C, F]
    ref = F_.group_norm(xt, 32, gamma, beta, eps=1e-5)
    ref = F_.silu(ref) if act else ref
    ref = ref.reshape(N, HW, C, F).permute(0, 3, 1, 2).reshape(N * F * HW, C)
    tol = 2e-5 if dt == torch.float32 else 1e-2
    assert rel_l2(y.float().cpu(), ref.cpu().numpy()) < tol
    assert rel_l2(y.float().cpu(), y2.float().cpu().numpy()) < tol


@pytest.mark.parametrize("name,N,F,qr,qg,kr,kg,win,heads", [
    ("spatial 1024", 2, 4, 4 * 1024, 1024, 4 * 1024, 1024, 1, 4), ("v<-a", 2, 16, 16 * 256, 256, 1600, 100, 1, 4),
    ("a<-v window 4", 1, 16, 1600, 100, 16 * 256, 256, 4, 6), ("ragged keys / queries", 2, 8, 8 * 77, 77, 8 * 50, 50, 3, 2),
    ("last group takes the remainder", 1, 16, 1610, 100, 16 * 64, 64, 8, 2), ("one short tile", 1, 16, 16 * 64, 64, 16 * 25, 25, 1, 8),
    ("v<-a ds2 full size", 1, 16, 16 * 1024, 1024, 6400, 400, 1, 4), ("600 queries, 7 resident tiles", 2, 4, 4 * 600, 600, 4 * 440, 440, 1, 2)])
def test_attn_dma_kernel_is_bitwise_the_mfma_kernel(ops, name, N, F, qr, qg, kr, kg, win, heads):
    """mmd_attn_fwd impl 4 (K / V tiles by buffer_load ... lds, V^T fragments by transposing LDS reads, one barrier per tile) against
    impl 2 (register-staged, transposing 2-byte LDS writes): same arithmetic in the same order -> bitwise equal; circular windows with
    a shift, key counts that are not multiples of 64 (zero-filled DMA rows + masking), ragged query tiles, the last group's
    remainder.  (impl 2 is the kernel test_ops_gpu.py pins against the oracle's attention.)  The training forward (mmd_attn_fwd_lse)
    runs the same kernel: same output."""
    ch = 64
    C = heads * ch
    g = torch.Generator(device="cuda").manual_seed(qr + kr)
    q = torch.randn(N * qr, 3 * C, device="cuda", generator=g).to(BF)
    kv = torch.randn(N * kr, 3 * C, device="cuda", generator=g).to(BF)
    for shift in (0, 5):
        sh = torch.tensor([shift], dtype=torch.int32, device="cuda")
        o2 = torch.zeros(N * qr, C, device="cuda", dtype=BF)
        o4 = torch.full((N * qr, C), 7.0, device="cuda", dtype=BF)
        ol = torch.full((N * qr, C), 7.0, device="cuda", dtype=BF)
        lse = torch.zeros(N * qr, heads, device="cuda")
        ops.attn(q, kv, o2, heads, ch, N, F, qr, qg, kr, kg, win, shift_dev=sh, impl=2)
        ops.attn(q, kv, o4, heads, ch, N, F, qr, qg, kr, kg, win, shift_dev=sh, impl=4)
        ops.attn_lse(q, kv, ol, lse, heads, ch, N, F, qr, qg, kr, kg, win, shift_dev=sh)
        torch.cuda.synchronize()
        assert torch.equal(o2, o4), f"{name} shift {shift}: rel-L2 {rel_l2(o4.float().cpu(), o2.float().cpu().numpy()):.3e}"
        assert torch.equal(o2, ol), f"{name} shift {shift} (lse forward)"


def test_graph_replays_are_bitwise_repeatable_under_two_stream_concurrency():
    """400 replays of the mid-size plan (video and audio chains concurrent inside the hipGraph) on the same inputs: every replay bitwise
    equal to the first.  Round 3 found 1 replay in ~130 off by 1e-2 rel-L2: gn_small_kernel's packed-fp32 accumulations went wrong in
    lanes 48-63 when its waves shared a SIMD with the other stream's GroupNorm+SiLU-in-the-loader GEMM; the library is built without
    packed fp32 since (build.py).  tools/determinism_*.py are the instruments that located it."""
    import random
    from helpers import flags, inputs
    from mm_diffusion import multimodal_script_util as msu
    from mm_diffusion.synth import synth_init_
    fl = flags("mid", use_fp16=True)
    model, _ = msu.create_model_and_diffusion(**fl)
    synth_init_(model)
    model.cuda().eval()
    v, a = inputs(fl, 2, 3)
    v, a, t = v.cuda(), a.cuda(), torch.tensor([17, 400]).cuda()

    def run():
        random.seed(5)
        with torch.no_grad():
            return model(v, a, t)
    ref = run()
    bad = [i for i in range(400) if not all(torch.equal(x, y) for x, y in zip(run(), ref))]
    model.release_engines()
    assert not bad, f"{len(bad)} of 400 replays differ from the first: {bad[:8]}"
