"""Row-strip 1x1 GEMM (conv_gemm tile 131, mmd_gemm.hip: conv1x1_strip_kernel) against the tiled main loops through the C-ABI.

The strip kernel keeps a wave's rows in registers as MFMA operands, applies GroupNorm once per strip and streams the weights
through LDS.  K order and epilogue arithmetic equal the tiled loops', so the OUTPUT is compared bitwise (tiles 128 / 129); the
fused-GroupNorm output against the tiled fused loader (same expressions) and against gn_apply + GEMM; the output statistics (own
summation order) against a float64 reduction of the stored values, to fp32 rounding."""
import pytest
import torch

from helpers import rel_l2

from mm_diffusion import ops as _ops

# MMD_GEMM_STRIP (ops._STRIP_MODE): "0" = tile 131 off, "base" = 1x1 convs at 128 / 256 / 384 channels (statistics up to 256), "pin" = all
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(_ops._STRIP_MODE == "0", reason="tile 131 switched off")]
full = pytest.mark.skipif(_ops._STRIP_MODE != "pin", reason="k=3 convs, K = 512 and one-fragment statistics need MMD_GEMM_STRIP=pin")
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    from mm_diffusion import ops as o
    return o


def _operands(M, Cin, Cout, res, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(M, Cin, device="cuda", generator=g).to(BF)
    w = (torch.randn(Cout, Cin, device="cuda", generator=g) * Cin ** -0.5).to(BF)
    b = torch.randn(Cout, device="cuda", generator=g)
    r = torch.randn(M, Cout, device="cuda", generator=g).to(BF) if res else None
    return x, w, b, r, g


@pytest.mark.parametrize("M,Cin,Cout,res", [
    (256 * 40, 128, 128, True),          # one column range per strip
    (65536, 256, 768, False),            # qkv conv at ds2, batch 4: two column ranges per strip
    (16384, 384, 1152, False),           # qkv conv at ds4 (one row fragment per wave, 32-channel chunks)
    (1600, 256, 256, True),              # audio ds4: ragged last strip (1600 = 6.25 x 256), deep column split
    (100, 128, 64, True),                # fewer rows than one wave pair
    (4096 + 32, 384, 96, True),          # ragged tail at 32-row granularity, three 32-channel chunks
    (25600, 256, 256, False),
    pytest.param(4096, 512, 1536, False, marks=full),            # ds8 qkv: K = 512, sixteen column ranges per strip
    pytest.param(4096 + 64, 512, 512, True, marks=full),
])
def test_strip_output_is_bitwise_the_tiled_output(ops, M, Cin, Cout, res):
    x, w, b, r, _ = _operands(M, Cin, Cout, res, M + Cin + Cout)
    y0 = ops.conv_gemm(x, w, b, residual=r, tile=129)
    assert torch.equal(y0, ops.conv_gemm(x, w, b, residual=r, tile=128))
    for _ in range(3):
        y1 = torch.full_like(y0, float("nan"))
        ops.conv_gemm(x, w, b, residual=r, tile=131, out=y1)
        assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
    y2 = ops.conv_gemm(x, w, None, residual=r, tile=131)            # no bias: the zero page
    assert torch.equal(y2, ops.conv_gemm(x, w, None, residual=r, tile=129))


@full
@pytest.mark.parametrize("N,F,HW,Cout,res,form", [(2, 16, 256, 128, False, "temporal"), (1, 5, 96, 64, True, "temporal"), (3, 1, 800, 128, True, "audio4"),
                                                  (2, 16, 64, 256, False, "temporal_d1")])
def test_strip_k3_convs_at_128_channels(ops, N, F, HW, Cout, res, form):
    """The k=3 temporal conv (D = (F, HW, 1), taps (+-1, 0, 0); also in the (N, F, HW) form) and the dilated audio conv (D = (L, 1, 1),
    taps (+-d, 0, 0)) at 128 channels: K = 384 is one strip; a 64-channel plane of the activation fragment comes from the row shifted
    by its tap, or from the zero row outside the clip.  Bitwise the tiled kernels' output, statistics to fp32 rounding."""
    Cin = 128
    if form == "temporal":
        M, taps, dims = N * F * HW, ops.TAPS_TEMPORAL, (F, HW, 1)
    elif form == "temporal_d1":
        M, taps, dims = N * F * HW, ops.TAPS_TEMPORAL_D1, (N, F, HW)
    else:
        M, taps, dims = N * HW, ops.taps_audio(4), (HW, 1, 1)
    g = torch.Generator(device="cuda").manual_seed(M + Cout)
    x = torch.randn(M, Cin, device="cuda", generator=g).to(BF)
    w = (torch.randn(Cout, 3 * Cin, device="cuda", generator=g) * (3 * Cin) ** -0.5).to(BF)
    b = torch.randn(Cout, device="cuda", generator=g)
    r = torch.randn(M, Cout, device="cuda", generator=g).to(BF) if res else None
    y0 = ops.conv_gemm(x, w, b, taps=taps, dims=dims, residual=r, tile=129)
    y1 = torch.full_like(y0, float("nan"))
    ops.conv_gemm(x, w, b, taps=taps, dims=dims, residual=r, tile=131, out=y1)
    assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
    if M % 64 == 0:
        rec = torch.full((M // 64, Cout // 4, 2), 7.0, device="cuda")           # one record per quad of channels (round 3)
        y2 = ops.conv_gemm(x, w, b, taps=taps, dims=dims, residual=r, tile=131, stats=rec)
        assert torch.equal(y0, y2)
        yf = y2.double().view(M // 64, 64, Cout // 4, 4)
        ref = torch.stack([yf.sum((1, 3)), (yf * yf).sum((1, 3))], dim=-1)
        assert float((rec.double() - ref).abs().max() / ref.abs().max()) < 2e-6


def test_strip_strided_views(ops):
    """Input, residual and output as column views of wider buffers (the skip concats of the up path)."""
    M, Cin, Cout = 2048 + 64, 256, 192
    g = torch.Generator(device="cuda").manual_seed(5)
    xw = torch.randn(M, Cin + 128, device="cuda", generator=g).to(BF)
    rw = torch.randn(M, Cout + 64, device="cuda", generator=g).to(BF)
    w = (torch.randn(Cout, Cin, device="cuda", generator=g) * Cin ** -0.5).to(BF)
    b = torch.randn(Cout, device="cuda", generator=g)
    x, r = xw[:, 64:64 + Cin], rw[:, 32:32 + Cout]
    yw0 = torch.zeros(M, Cout + 16, device="cuda", dtype=BF)
    yw1 = torch.zeros(M, Cout + 16, device="cuda", dtype=BF)
    ops.conv_gemm(x, w, b, residual=r, tile=129, out=yw0[:, 8:8 + Cout])
    ops.conv_gemm(x, w, b, residual=r, tile=131, out=yw1[:, 8:8 + Cout])
    assert torch.equal(yw0, yw1)


@pytest.mark.parametrize("S,Tn,Cin,Cout,act,res", [
    (4, 1024, 256, 768, False, False),   # spatial-attention qkv: per-frame slices, normalised once per strip for all 768 columns
    (2, 4096, 128, 128, True, True),     # ResBlock tail at ds1: norm -> SiLU -> out conv -> + skip
    (3, 400, 256, 256, True, True),      # audio slices that are not a multiple of the strip: a strip straddles two samples
    (6, 256, 384, 1152, False, False),   # ds4 qkv, slices of exactly one 128-row block pair
    (5, 320, 128, 64, True, False),
])
def test_strip_fused_groupnorm(ops, S, Tn, Cin, Cout, act, res):
    M = S * Tn
    x, w, b, r, g = _operands(M, Cin, Cout, res, S * Tn + Cout)
    x = (x.float() * 1.4 + 0.3).to(BF)
    gamma, beta = 1 + 0.1 * torch.randn(Cin, device="cuda", generator=g), torch.randn(Cin, device="cuda", generator=g)
    film = torch.randn(S, 2 * Cin, device="cuda", generator=g) * 0.3
    geom = ops.Geom.per_sample(S, Tn)
    ga, gb = ops.gn_stats(x, gamma, beta, geom, film=film)
    y = torch.full((M, Cout), float("nan"), device="cuda", dtype=BF)
    ops.gn_conv1x1(x, ga, gb, geom, act, w, b, residual=r, tile=131, out=y)
    # (a) unfused path: normalised tensor in HBM, then the plain GEMM
    xn = ops.gn_apply(x, ga, gb, geom, act=act)
    y_unfused = ops.conv_gemm(xn, w, b, residual=r, tile=129)
    e_unf = rel_l2(y.float().cpu(), y_unfused.float().cpu().numpy())
    # same expressions, same rounding point: fusing is a pure speed choice (ops.gn_fusable makes it per launch size)
    assert torch.equal(y.view(torch.int16), y_unfused.view(torch.int16))
    # (b) the tiled fused loader where it applies (same expressions in the loader)
    msg = ""
    if Cin <= 256 and Cout <= 256:
        y_tiled = ops.gn_conv1x1(x, ga, gb, geom, act, w, b, residual=r, tile=128)
        same = torch.equal(y, y_tiled)
        e_t = rel_l2(y.float().cpu(), y_tiled.float().cpu().numpy())
        msg = f", vs tiled fused loader: rel-L2 {e_t:.2e} bitwise={same}"
        assert e_t < 2e-3
    # (c) float64 reference on the stored inputs
    sl = torch.arange(M, device="cuda") // Tn
    xr = x.double() * ga.double()[sl] + gb.double()[sl]
    if act:
        xr = xr * torch.sigmoid(xr)
    ref = xr.to(BF).double() @ w.double().t() + b.double()
    if r is not None:
        ref = ref + r.double()
    e_ref = rel_l2(y.float().cpu(), ref.float().cpu().numpy())
    print(f"strip fused GN S={S} Tn={Tn} {Cin}->{Cout}: vs fp64 {e_ref:.2e}, vs gn_apply + GEMM {e_unf:.2e}{msg}")
    assert e_ref < 6e-3 and e_unf < 4e-3


@pytest.mark.parametrize("M,Cin,Cout,res,gn", [(4096, 128, 128, True, True), (2048 + 64, 256, 320, False, False), (8192, 256, 256, True, True),
                                               pytest.param(16384, 384, 384, True, False, marks=full), pytest.param(4096 + 128, 512, 512, True, False, marks=full),
                                               pytest.param(2048, 384, 96, False, True, marks=full)])
def test_strip_output_statistics(ops, M, Cin, Cout, res, gn):
    """Per (64-row record, quad of 4 columns) sum / sum of squares of the values as stored, written into a quad slice of a wider record
    buffer; emitting them does not change the output."""
    x, w, b, r, g = _operands(M, Cin, Cout, res, M + 7)
    Q = Cout // 4
    wide = torch.full((M // 64, Q + 10, 2), 7.0, device="cuda")
    view = wide[:, 6:6 + Q, :]
    if gn:
        S = 2
        geom = ops.Geom.per_sample(S, M // S)
        ga, gb = ops.gn_stats(x, torch.ones(Cin, device="cuda"), torch.zeros(Cin, device="cuda"), geom)
        y0 = ops.gn_conv1x1(x, ga, gb, geom, True, w, b, residual=r, tile=131)
        y1 = ops.gn_conv1x1(x, ga, gb, geom, True, w, b, residual=r, tile=131, stats=view)
    else:
        y0 = ops.conv_gemm(x, w, b, residual=r, tile=131)
        y1 = ops.conv_gemm(x, w, b, residual=r, tile=131, stats=view)
    assert torch.equal(y0, y1)
    yf = y1.double().view(M // 64, 64, Q, 4)
    ref = torch.stack([yf.sum((1, 3)), (yf * yf).sum((1, 3))], dim=-1)
    assert float((view.double() - ref).abs().max() / ref.abs().max()) < 2e-6
    assert float((wide[:, :6] - 7).abs().max()) == 0 and float((wide[:, 6 + Q:] - 7).abs().max()) == 0
    # the next norm's affine from these records == the statistics pass over the stored output
    geom2 = ops.Geom.per_sample(2, M // 2) if (M // 2) % 64 == 0 and Cout % 128 == 0 else None
    if geom2 is not None:
        g2, b2 = torch.randn(Cout, device="cuda", generator=g), torch.randn(Cout, device="cuda", generator=g)
        an, bn = ops.gn_finalize_stats(view, g2, b2, geom2)
        ar, br = ops.gn_stats(y1, g2, b2, geom2)
        assert rel_l2(an.cpu(), ar.cpu().numpy()) < 1e-5 and rel_l2(bn.cpu(), br.cpu().numpy()) < 1e-5


def test_strip_rows_do_not_depend_on_the_batch(ops):
    """The rows (and statistics records) of sample 0 computed in a batch of 4 and alone: other grid, other column split, same bits."""
    Tn, Cin, Cout = 1024 * 4, 256, 768
    x, w, b, _, g = _operands(4 * Tn, Cin, Cout, False, 3)
    geom4, geom1 = ops.Geom.per_sample(4, Tn), ops.Geom.per_sample(1, Tn)
    gamma, beta = torch.randn(Cin, device="cuda", generator=g), torch.randn(Cin, device="cuda", generator=g)
    a4, b4 = ops.gn_stats(x, gamma, beta, geom4)
    y4 = ops.gn_conv1x1(x, a4, b4, geom4, False, w, b, tile=131)
    y1 = ops.gn_conv1x1(x[:Tn], a4[:1].contiguous(), b4[:1].contiguous(), geom1, False, w, b, tile=131)
    assert torch.equal(y4[:Tn], y1)
    w2 = w[:256].contiguous()
    r4 = torch.zeros(4 * Tn // 64, 64, 2, device="cuda")
    r1 = torch.zeros(Tn // 64, 64, 2, device="cuda")
    z4 = ops.conv_gemm(x, w2, b[:256].contiguous(), tile=131, stats=r4)
    z1 = ops.conv_gemm(x[:Tn], w2, b[:256].contiguous(), tile=131, stats=r1)
    assert torch.equal(z4[:Tn], z1) and torch.equal(r4[:Tn // 64], r1)


@full
def test_strip_is_pinned_by_layer_geometry(ops):
    x = torch.zeros(512, 256, device="cuda", dtype=BF)
    assert ops.strip_tile_pinned(x, 768) and ops.strip_tile_pinned(x, 256, stats=torch.zeros(8, 64, 2))
    assert not ops.strip_tile_pinned(x.float(), 768)                               # fp32 mode keeps the exact-fp32 tiles
    assert ops.strip_tile_pinned(torch.zeros(512, 512, device="cuda", dtype=BF), 512)
    assert not ops.strip_tile_pinned(torch.zeros(512, 640, device="cuda", dtype=BF), 512)
    x3 = torch.zeros(512, 384, device="cuda", dtype=BF)
    assert ops.strip_tile_pinned(x3, 1152) and ops.strip_tile_pinned(x3, 384, stats=torch.zeros(8, 96, 2))
    x1 = torch.zeros(512, 128, device="cuda", dtype=BF)
    assert ops.strip_tile_pinned(x1, 128, taps=ops.TAPS_TEMPORAL) and not ops.strip_tile_pinned(x, 256, taps=ops.TAPS_TEMPORAL)   # K = 768
    assert not ops.strip_tile_pinned(x, 768, geom=ops.Geom.per_sample(4, 128))    # slices shorter than a strip: gn_apply + GEMM
