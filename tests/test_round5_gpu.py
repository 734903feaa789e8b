"""Round-5 kernels and launch-plan changes (through the C ABI): the resample with a statistics epilogue, the head as GEMM + gather, the
out layers of the up ResBlocks at the input resolution, the serialised cross-attention pair.

The kernel tests compare with fp32 torch restatements of the reference operators (unet:133-208 resampling, nn.py:16-33 GroupNorm32,
unet:1003-1012 head); the plan tests compare the engine built with a switch on against the same engine with it off AND against the
reference-generated fixtures (tests/golden)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F_

from helpers import rel_l2

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    from mm_diffusion import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16)


# --------------------------------------------------------------------------- resample + statistics epilogue
@pytest.mark.parametrize("NF,H,W,fh,fw,mode,C,ld_extra,c0", [
    (8, 16, 16, 2, 2, 1, 128, 0, 0),        # video upsample, 16 vectors per row (16 row groups)
    (8, 16, 16, 2, 2, 0, 256, 0, 0),        # video avg-pool
    (4, 8, 8, 2, 2, 1, 384, 0, 0),          # 48 vectors per row: padded to 64, a quarter of the threads idle
    (4, 8, 8, 2, 2, 1, 512, 256, 256),      # output = the right-hand column slice of a wider buffer (and of its record buffer)
    (2, 1, 1024, 1, 4, 0, 128, 0, 0),       # audio avg-pool by 4
    (2, 1, 256, 1, 4, 1, 256, 128, 0),      # audio upsample into the left-hand slice of a wider buffer
    (70, 16, 16, 2, 2, 1, 128, 0, 0),       # 1120 records (the grid-stride path, > 8192 records, is the test below)
])
def test_resample_stats(ops, NF, H, W, fh, fw, mode, C, ld_extra, c0):
    x = rnd(NF * H * W, C, seed=3).cuda()
    Ho, Wo = (H // fh, W // fw) if mode == 0 else (H * fh, W * fw)
    rows = NF * Ho * Wo
    assert rows % 64 == 0
    Ct = C + ld_extra
    buf = torch.zeros(rows, Ct, dtype=torch.bfloat16, device="cuda")
    rec = torch.full((rows // 64, Ct // 4, 2), float("nan"), dtype=torch.float32, device="cuda")
    out = buf[:, c0:c0 + C]
    ops.resample(x, out, NF, H, W, fh, fw, mode, stats=rec[:, c0 // 4:(c0 + C) // 4, :])
    plain = torch.zeros(rows, C, dtype=torch.bfloat16, device="cuda")
    ops.resample(x, plain, NF, H, W, fh, fw, mode)
    assert torch.equal(out, plain), "the statistics variant must store exactly what mmd_resample stores"
    if ld_extra:
        other = torch.cat([buf[:, :c0], buf[:, c0 + C:]], 1)
        assert float(other.float().abs().max()) == 0.0, "columns outside the slice were written"
    # records against fp64 sums of the STORED values
    v = out.double().cpu().view(rows // 64, 64, C // 4, 4)
    ref = torch.stack([v.sum((1, 3)), (v * v).sum((1, 3))], -1)
    got = rec[:, c0 // 4:(c0 + C) // 4, :].double().cpu()
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max() / ref.abs().max()) < 2e-6
    if ld_extra:      # records of the other columns untouched
        mask = torch.ones(Ct // 4, dtype=torch.bool)
        mask[c0 // 4:(c0 + C) // 4] = False
        assert torch.isnan(rec[:, mask.cuda(), :]).all()
    # torch restatement of the operator itself (unet:133-208) on the bf16-rounded input
    xr = x.float().cpu().view(NF, H, W, C).permute(0, 3, 1, 2)
    yr = F_.avg_pool2d(xr, (fh, fw)) if mode == 0 else F_.interpolate(xr, scale_factor=(fh, fw), mode="nearest")
    assert rel_l2(out.float().cpu(), yr.permute(0, 2, 3, 1).reshape(rows, C)) < 4e-3
    # bitwise repeatable
    rec2 = torch.zeros_like(rec)
    ops.resample(x, out, NF, H, W, fh, fw, mode, stats=rec2[:, c0 // 4:(c0 + C) // 4, :])
    assert torch.equal(rec2[:, c0 // 4:(c0 + C) // 4, :], rec[:, c0 // 4:(c0 + C) // 4, :])


def test_resample_stats_many_records_and_finalize(ops):
    """More records than the launch has blocks (grid-stride loop), and the records drive mmd_gn_finalize_stats to the same fused affine
    as the statistics pass over the resampled tensor."""
    N, F, H, W, C = 3, 16, 64, 64, 128                  # upsample to 128 x 128: 12288 records on a grid of 8192 blocks
    x = rnd(N * F * H * W, C, seed=5).cuda()
    rows = N * F * 4 * H * W
    out = torch.empty(rows, C, dtype=torch.bfloat16, device="cuda")
    rec = torch.zeros(rows // 64, C // 4, 2, dtype=torch.float32, device="cuda")
    ops.resample(x, out, N * F, H, W, 2, 2, 1, stats=rec)
    g = ops.Geom.per_sample(N, rows // N)
    gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    a1, b1 = ops.gn_finalize_stats(rec, gamma, beta, g)
    a2, b2 = ops.gn_stats(out, gamma, beta, g)
    assert rel_l2(a1.cpu(), a2.cpu()) < 2e-5 and rel_l2(b1.cpu(), b2.cpu()) < 2e-5
    # ... which are the statistics of the tensor BEFORE the upsample (what the up ResBlocks rely on)
    a3, b3 = ops.gn_stats(x, gamma, beta, ops.Geom.per_sample(N, rows // N // 4))
    assert rel_l2(a1.cpu(), a3.cpu()) < 2e-5 and rel_l2(b1.cpu(), b3.cpu()) < 2e-5


def test_resample_stats_rejects(ops):
    from mm_diffusion._hip import MMDError
    x = rnd(4 * 8 * 8, 128).cuda()
    out = torch.empty(4 * 16 * 16, 128, dtype=torch.bfloat16, device="cuda")
    rec = torch.zeros(16, 32, 2, dtype=torch.float32, device="cuda")
    with pytest.raises(MMDError):
        ops.resample(x.float(), out.float(), 4, 8, 8, 2, 2, 1, stats=rec)             # fp32 rows
    with pytest.raises(MMDError):
        ops.resample(x, out, 4, 8, 8, 2, 2, 1, scale=0.25, stats=rec)                   # scaled (backward form)
    with pytest.raises(MMDError):
        ops.resample(x, out, 4, 8, 8, 2, 2, 1, stats=rec[:, 1:, :])                     # wrong record shape
    x2 = rnd(3 * 4 * 4, 128).cuda()                                                      # 48 -> 12 output rows: not whole records
    with pytest.raises(MMDError):
        ops.resample(x2, torch.empty(12, 128, dtype=torch.bfloat16, device="cuda"), 3, 4, 4, 2, 2, 0,
                     stats=torch.zeros(0, 32, 2, dtype=torch.float32, device="cuda"))


# --------------------------------------------------------------------------- head: GEMM + gather
def _head_ref(x_rows, a, b, S, w, bias, N, F, H, W):
    """fp64 restatement: SiLU(x a + b) rounded to bf16 (where gn_apply stores it), Conv3d 3x3x3 with the fp32 weights (unet:1003-1012)."""
    C = x_rows.shape[1]
    rows_per = x_rows.shape[0] // S
    xn = x_rows.float().view(S, rows_per, C) * a.view(S, 1, C) + b.view(S, 1, C)
    xn = (xn * torch.sigmoid(xn)).to(torch.bfloat16).double().view(N, F, H, W, C).permute(0, 4, 1, 2, 3)
    return F_.conv3d(xn, w.double(), bias.double(), padding=1).permute(0, 2, 1, 3, 4)      # [N, F, Co, H, W]


@pytest.mark.parametrize("N,F,H,W,Co", [(2, 4, 8, 8, 3), (1, 16, 16, 16, 3), (2, 2, 8, 16, 1), (1, 4, 8, 8, 2)])
def test_head_gemm_gather(ops, N, F, H, W, Co):
    C = 128
    M = N * F * H * W
    x = rnd(M, C, seed=11).cuda()
    g = torch.Generator().manual_seed(12)
    a = (torch.rand(N, C, generator=g) + 0.5)
    b = torch.randn(N, C, generator=g) * 0.3
    w = torch.randn(Co, C, 3, 3, 3, generator=g) * (27 * C) ** -0.5
    bias = torch.randn(Co, generator=g)
    geom = ops.Geom.per_sample(N, F * H * W)
    wp = ops.pack_edge_weight(w).cuda()
    assert ops.head_gemm_ok(x, wp, geom)
    NO = 27 * Co
    P = torch.full((NO, M), float("nan"), dtype=torch.float32, device="cuda")
    ops.head_gemm(x, a.cuda(), b.cuda(), geom, True, ops.head_gemm_pack(wp), P, NO)
    y = torch.full((N, F, Co, H, W), float("nan"), dtype=torch.float32, device="cuda")
    ops.head_gather(P, bias.cuda(), y, N, F, H, W, Co, ops.TAPS_3D)
    ref = _head_ref(x.cpu(), a, b, N, w, bias, N, F, H, W)
    assert torch.isfinite(y).all()
    err = rel_l2(y.cpu(), ref)
    # fp32 accumulation of bf16 x (hi + lo) products: the (hi, lo) weight pair keeps 2^-17 of the fp32 weights; what is left is SiLU's
    # fast exp / rcp flipping a bf16 rounding of the activation here and there
    assert err < 1.5e-3, err
    # ... and the path it replaces (gn_apply + the direct kernel, fp32 weights) sits at the same distance
    hv = ops.gn_apply(x, a.cuda(), b.cuda(), geom, act=True)
    y0 = torch.zeros_like(y)
    ops.head_conv(hv, wp, bias.cuda(), y0, N, F, H, W, ops.TAPS_3D)
    assert rel_l2(y.cpu(), y0.cpu()) < 2e-5, rel_l2(y.cpu(), y0.cpu())


def test_head_gemm_full_size_batch_rows(ops):
    """The head of the full-size model (batch 2 of it): per-sample affine tables, row groups crossing block boundaries; sample 1 alone
    gives the same bits (the kernels are row-local)."""
    N, F, H, W, C, Co = 2, 16, 64, 64, 128, 3
    M = N * F * H * W
    x = rnd(M, C, seed=21).cuda()
    g = torch.Generator().manual_seed(22)
    a, b = (torch.rand(N, C, generator=g) + 0.5).cuda(), (torch.randn(N, C, generator=g) * 0.3).cuda()
    wp = ops.pack_edge_weight(torch.randn(Co, C, 3, 3, 3, generator=g) * (27 * C) ** -0.5).cuda()
    bias = torch.randn(Co, generator=g).cuda()
    wimg = ops.head_gemm_pack(wp)

    def run(xs, a_, b_, n):
        P = torch.empty(81, xs.shape[0], dtype=torch.float32, device="cuda")
        ops.head_gemm(xs, a_, b_, ops.Geom.per_sample(n, F * H * W), True, wimg, P, 81)
        y = torch.empty(n, F, Co, H, W, dtype=torch.float32, device="cuda")
        ops.head_gather(P, bias, y, n, F, H, W, Co, ops.TAPS_3D)
        return y

    y = run(x, a, b, N)
    y1 = run(x[M // 2:].contiguous(), a[1:].contiguous(), b[1:].contiguous(), 1)
    assert torch.equal(y[1:], y1)
    hv = ops.gn_apply(x, a, b, ops.Geom.per_sample(N, F * H * W), act=True)
    y0 = torch.zeros_like(y)
    ops.head_conv(hv, wp, bias, y0, N, F, H, W, ops.TAPS_3D)
    assert rel_l2(y.cpu(), y0.cpu()) < 2e-5


# --------------------------------------------------------------------------- launch-plan switches: same model, switch on / off
# (the switches are read once per process, so each arm is its own interpreter)
_SWITCH_SCRIPT = r"""
import os, sys
for p in ({root!r}, os.path.join({root!r}, "mm-diffusion_amd"), os.path.join({root!r}, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from helpers import flags, synth_sd, inputs, gold
from mm_diffusion import multimodal_script_util as msu, logger
logger.set_quiet(True)
cfg, keyset, tag, bf16 = {cfg!r}, {keyset!r}, {tag!r}, {bf16!r}
g = gold(tag + "_forward")
f = flags(cfg, use_fp16=bf16)
model, _ = msu.create_model_and_diffusion(**f)
model.load_state_dict(synth_sd(keyset)); model.cuda().eval()
video, audio = inputs(f, int(g["B"]), int(g["seed"]))
outs = []
for rep in range(2):
    it = iter(int(s) for s in g["shifts"])
    model.shift_source = lambda lo, hi: next(it)
    with torch.no_grad():
        outs.append(model(video.cuda(), audio.cuda(), torch.from_numpy(g["t"]).cuda()))
assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "graph replay is not bitwise repeatable"
eng = next(iter(model._engines.values()))
names = [e[2] for e in eng.plan if e[0] is not None]
np.savez({out!r}, vo=outs[0][0].float().cpu().numpy(), ao=outs[0][1].float().cpu().numpy(), nlaunch=len(names), names=np.array(names),
         nsync=len([e for e in eng.plan if e[0] is None]))
"""
_OFF = {"MMD_UP_LOWRES": "0", "MMD_RESAMPLE_STATS": "0", "MMD_CROSS_SERIAL": "0", "MMD_HEAD_GEMM": "0"}


def _run_switch(tmp_path, cfg, keyset, tag, bf16, env, arm):
    out = str(tmp_path / f"{arm}.npz")
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", _SWITCH_SCRIPT.format(root=ROOT, cfg=cfg, keyset=keyset, tag=tag, bf16=bf16, out=out)], env=e,
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(out)


@pytest.mark.parametrize("cfg,keyset,tag,bf16", [("mid", "tiny", "mid", True), ("mid", "tiny", "mid", False), ("full", "full", "full", True)])
def test_round5_plan_switches_against_reference_fixture(tmp_path, cfg, keyset, tag, bf16):
    """A forward on the reference fixture with every round-5 plan change ON (the default) and OFF: both inside the fixture bound, close
    to each other, each bitwise repeatable, and the new plan has fewer launches and fewer cross-stream waits."""
    on = {k: "1" for k in _OFF}
    a = _run_switch(tmp_path, cfg, keyset, tag, bf16, on, "on")
    b = _run_switch(tmp_path, cfg, keyset, tag, bf16, _OFF, "off")
    g = np.load(os.path.join(ROOT, "tests", "golden", tag + "_forward.npz"))
    bound = 2.5e-2 if bf16 else 2e-5
    for k, ref in (("vo", g["video_out"]), ("ao", g["audio_out"])):
        ea, eb, ab = rel_l2(a[k], ref), rel_l2(b[k], ref), rel_l2(a[k], b[k])
        print(f"{tag} bf16={bf16} {k}: on {ea:.3e} off {eb:.3e} on-vs-off {ab:.3e}")
        assert ea < bound and eb < bound
        # bf16: the two plans are two independent roundings of the same network, each ~1.6e-2 from the fp32 reference on this small model,
        # so they sit up to ~sqrt(2) x that apart (1.49e-2 ... 1.54e-2 measured, depending on which bitwise-different norm kernels a plan uses)
        assert ab < (2.0e-2 if bf16 else 5e-6)
    assert int(a["nlaunch"]) < int(b["nlaunch"]) and int(a["nsync"]) < int(b["nsync"])
    na, nb = set(a["names"].tolist()), set(b["names"].tolist())
    if bf16:
        assert "mmd_resample_stats" in na and "mmd_resample_stats" not in nb
    if tag == "full":
        assert "mmd_head_gemm" in na and "mmd_head_gather" in na and "mmd_head_conv" in nb
        print("launches per forward:", int(a["nlaunch"]), "(was", int(b["nlaunch"]), ")")


# --------------------------------------------------------------------------- cross-stream ordering: alternating inputs on one engine
@pytest.mark.parametrize("cfg,B,iters", [("mid", 1, 200), ("mid", 2, 100), ("full", 1, 40)])
def test_alternating_inputs_replay_bitwise(cfg, B, iters):
    """tools/alternating_inputs_stress.py: two different inputs fed alternately to ONE engine (graph replay of the two-stream plan); every
    output must be bitwise its first occurrence.  Replaying the same input cannot see a launch that runs ahead of its producer on the other
    stream (it reads the previous replay's identical data); with alternating inputs the stale data is the other input's."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("alt_stress", os.path.join(ROOT, "tools", "alternating_inputs_stress.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    nbad, worst = m.run(cfg, B, iters, verbose=True)
    assert nbad == 0, f"{nbad} of {iters} alternating replays differ (worst rel-L2 {worst:.2e})"
