"""CPU restatements of the round-5 plan rewrites, each checked against the oracle's own functions (oracle/unet_ref.py, itself pinned to
the reference by tests/test_oracle_golden.py): the algebra the engine relies on, independent of any kernel.

  * up ResBlocks: everything behind the nearest upsample is pointwise, so the block at the INPUT resolution followed by one upsample
    equals the reference order (engine.py `_UP_LOWRES`; reference unet:441-448, 457-476)
  * the video head 3x3x3 conv 128 -> 3 as product planes P[tap, co][m] = W . act(norm(x[m])) plus a 27-tap gather with the index
    arithmetic of csrc/mmd_misc.hip head_gather_kernel (unet:1003-1012), and the (hi, lo) bf16 weight image of ops.head_gemm_pack
  * GroupNorm statistics from 64-row x 4-channel (sum, sum of squares) records - what mmd_resample_stats and the GEMM epilogues emit and
    mmd_gn_finalize_stats folds (nn:16-33)
"""
import numpy as np
import torch
import torch.nn.functional as F_

from helpers import flags, rel_l2, synth_sd
from oracle import unet_ref as uref


def _up_layers(cfg):
    ins, mid, outs = uref.build_arch(cfg)
    return [l for blk in outs for l in blk if l["kind"] == "res" and l["up"]]


def test_up_resblock_at_input_resolution_then_one_upsample():
    fl = flags("tiny")
    cfg = uref.parse_cfg(fl)
    sd = {k: v.float() for k, v in synth_sd("tiny").items()}
    ups = _up_layers(cfg)
    assert len(ups) == 3
    g = torch.Generator().manual_seed(11)
    for layer in ups:
        C = layer["cin"]
        video = torch.randn(2, C, 8, 4, 4, generator=g)
        audio = torch.randn(2, C, 32, generator=g)
        emb = torch.randn(2, sd[layer["prefix"] + ".emb_layers.1.weight"].shape[1], generator=g)
        ref_v, ref_a = uref.res_block(video, audio, emb, sd, layer, cfg)
        low = dict(layer, up=False)
        lv, la = uref.res_block(video, audio, emb, sd, low, cfg)
        got_v = lv.repeat_interleave(2, dim=3).repeat_interleave(2, dim=4)
        got_a = la.repeat_interleave(4, dim=2)
        assert got_v.shape == ref_v.shape and got_a.shape == ref_a.shape
        # equal up to the rounding of the GroupNorm sums (4x / 4x as many equal terms in the reference order)
        assert rel_l2(got_v, ref_v.numpy()) < 2e-6 and rel_l2(got_a, ref_a.numpy()) < 2e-6


def _gather_model(P, bias, N, F, H, W, Co, taps):
    """head_gather_kernel, vectorised: thread m = ((n F + f) H + h) W + w; a tap outside the frame reads the centre element times zero."""
    M = N * F * H * W
    m = np.arange(M)
    w, h, f, n = m % W, (m // W) % H, (m // (W * H)) % F, m // (W * H * F)
    acc = np.tile(bias.astype(np.float32), (M, 1))
    for t, (df, dh, dw) in enumerate(taps):
        ok = (f + df >= 0) & (f + df < F) & (h + dh >= 0) & (h + dh < H) & (w + dw >= 0) & (w + dw < W)
        src = np.where(ok, m + df * H * W + dh * W + dw, m)
        for c in range(Co):
            acc[:, c] += ok.astype(np.float32) * P[t * Co + c, src]
    y = np.zeros((N, F, Co, H, W), np.float32)
    y[n[:, None], f[:, None], np.arange(Co)[None, :], h[:, None], w[:, None]] = acc
    return y


def test_head_conv_as_product_planes_and_gather():
    from mm_diffusion import ops
    N, F, C, H, W, Co = 2, 4, 128, 6, 5, 3
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, C, F, H, W, generator=g)                     # b c f h w, as the oracle's head sees it
    gw, gb = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    wt = torch.randn(Co, C, 3, 3, 3, generator=g) / np.sqrt(27 * C)
    bias = 0.1 * torch.randn(Co, generator=g)
    a = F_.silu(uref.group_norm(x, gw, gb))
    ref = F_.conv3d(a, wt, bias, padding=1).permute(0, 2, 1, 3, 4)   # -> [N, F, Co, H, W], the model's output layout
    rows = a.permute(0, 2, 3, 4, 1).reshape(N * F * H * W, C)        # row m = ((n F + f) H + h) W + w
    wp = ops.pack_edge_weight(wt)                                    # [27, Cin, Co]
    assert len(ops.TAPS_3D) == 27 and tuple(ops.TAPS_3D[13]) == (0, 0, 0)
    Wm = wp.permute(0, 2, 1).reshape(27 * Co, C)                     # W[tap Co + co][ci]
    P = (Wm.double() @ rows.double().T).float().numpy()
    got = _gather_model(P, bias.numpy(), N, F, H, W, Co, [tuple(t) for t in ops.TAPS_3D])
    assert rel_l2(torch.from_numpy(got), ref.numpy()) < 2e-6
    # the (hi, lo) bf16 weight image: [hl][ob][cg][half][l31][8], lane (l31, half) of (ob, cg) = W[32 ob + l31][16 cg + 8 half .. + 8]
    full = torch.zeros(96, C)
    full[:27 * Co] = Wm
    hi = full.to(torch.bfloat16)
    lo = (full - hi.float()).to(torch.bfloat16)
    img = torch.stack([hi, lo]).view(2, 3, 32, C // 16, 2, 8).permute(0, 1, 3, 4, 2, 5).contiguous()
    assert img.numel() * 2 == 2 * 3 * (C // 16) * 1024                # mmd_head_gemm_weight_bytes(128)
    for ob, cg, half, l31 in ((0, 0, 0, 0), (2, 7, 1, 31), (1, 3, 0, 17)):
        assert torch.equal(img[0, ob, cg, half, l31], hi[32 * ob + l31, 16 * cg + 8 * half: 16 * cg + 8 * half + 8])
        assert torch.equal(img[1, ob, cg, half, l31], lo[32 * ob + l31, 16 * cg + 8 * half: 16 * cg + 8 * half + 8])
    # two bf16 terms carry the fp32 weight to 2^-16 of its magnitude (the GEMM runs both against the same bf16 activations)
    err = (hi.float() + lo.float() - full).abs().max() / full.abs().max()
    assert float(err) < 2.0 ** -15


def test_groupnorm_from_quad_records_of_an_upsampled_tensor():
    """Records [rows / 64, C / 4, (sum, sumsq)] of the stored tensor -> per (sample, group) mean / rstd -> y = x a + b equals the
    oracle's GroupNorm; and the records of a nearest-upsampled tensor give the statistics of the tensor it was upsampled from."""
    N, F, C, H, W = 2, 4, 128, 4, 4
    g = torch.Generator().manual_seed(9)
    low = torch.randn(N, C, F, H, W, generator=g) * 2 + 0.5
    up = low.repeat_interleave(2, dim=3).repeat_interleave(2, dim=4)
    gw, gb = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)

    def affine_from_records(x):
        rows = x.permute(0, 2, 3, 4, 1).reshape(-1, C)               # channels-last rows, one sample = rows / N consecutive rows
        per = rows.shape[0] // N
        assert per % 64 == 0
        v = rows.double().view(-1, 64, C // 4, 4)
        rec = torch.stack([v.sum((1, 3)), (v * v).sum((1, 3))], -1)   # [rows / 64, C / 4, 2]
        cpg = C // 32
        assert cpg % 4 == 0                                           # groups are whole quads
        r = rec.view(N, per // 64, 32, cpg // 4, 2).sum((1, 3))       # fold a sample's records and a group's quads
        cnt = per * cpg
        mean = r[..., 0] / cnt
        var = r[..., 1] / cnt - mean * mean
        rstd = (var + 1e-5).rsqrt()
        a = gw.double().view(1, 32, cpg) * rstd.unsqueeze(-1)
        b = gb.double().view(1, 32, cpg) - mean.unsqueeze(-1) * a
        return a.reshape(N, C), b.reshape(N, C), mean, rstd

    a, b, mean_up, rstd_up = affine_from_records(up)
    y = up.double() * a.view(N, C, 1, 1, 1) + b.view(N, C, 1, 1, 1)
    assert rel_l2(y.float(), uref.group_norm(up, gw, gb).numpy()) < 2e-6
    _, _, mean_low, rstd_low = affine_from_records(low)
    np.testing.assert_allclose(mean_up.numpy(), mean_low.numpy(), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(rstd_up.numpy(), rstd_low.numpy(), rtol=1e-12)
