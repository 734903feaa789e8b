#!/usr/bin/env python3
"""mmd_tattn_block (the fused temporal-attention block) against the four launches it replaces - gn_small, qkv 1x1 conv, attn_small,
proj_out 1x1 conv + residual with statistics - on the ds2 shape of the Landscape model at batch 4 (65536 rows x 256 channels).
HIP-event times, 20 launches each, interleaved rounds."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-diffusion_amd"))
import torch  # noqa: E402
from mm_diffusion import _hip as H, ops  # noqa: E402


def timed(fn, n=20):
    ev = [ctypes.c_void_p(), ctypes.c_void_p()]
    for e in ev:
        H.call("mmd_event_create", ctypes.byref(e))
    st = H.stream_handle()
    for _ in range(3):
        fn()
    H.call("mmd_event_record", ev[0], st)
    for _ in range(n):
        fn()
    H.call("mmd_event_record", ev[1], st)
    ms = ctypes.c_float()
    H.call("mmd_event_elapsed_ms", ev[0], ev[1], ctypes.byref(ms))
    for e in ev:
        H.lib().mmd_event_destroy(e)
    return ms.value / n * 1000


def main():
    heads, F = 4, 16
    for N, HW, C in ((4, 1024, 256), (1, 1024, 256), (4, 256, 256)):      # (the 384 / 512-channel instances were removed in round 5)
        M = N * F * HW
        g = torch.Generator(device="cuda").manual_seed(0)
        x = torch.randn(M, C, device="cuda", generator=g).to(torch.bfloat16)
        wqkv = (torch.randn(3 * C, C, device="cuda", generator=g) * C ** -0.5).to(torch.bfloat16)
        wproj = (torch.randn(C, C, device="cuda", generator=g) * C ** -0.5).to(torch.bfloat16)
        bqkv, bproj = torch.randn(3 * C, device="cuda", generator=g), torch.randn(C, device="cuda", generator=g)
        gamma, beta = torch.rand(C, device="cuda", generator=g) + 0.5, torch.randn(C, device="cuda", generator=g) * 0.3
        wf = ops.tattn_pack(wqkv, wproj)
        geom = ops.Geom.temporal(N, F, HW)
        n1, att, y0, y1 = (torch.empty(M, C, device="cuda", dtype=torch.bfloat16) for _ in range(4))
        qkv = torch.empty(M, 3 * C, device="cuda", dtype=torch.bfloat16)
        rec = torch.zeros(M // 64, C // 4, 2, device="cuda")

        def four():
            ops.gn_small(x, gamma, beta, geom, act=False, out=n1)
            ops.conv_gemm(n1, wqkv, bqkv, out=qkv)
            ops.attn_small(qkv, att, C, heads, geom)
            ops.conv_gemm(att, wproj, bproj, residual=x, out=y0, stats=rec)

        def fused():
            ops.tattn_block(x, wf, bqkv, bproj, gamma, beta, heads, N, F, HW, out=y1, stats=rec)

        att_s = torch.randn(M, C, device="cuda", generator=g).to(torch.bfloat16)
        wpre, bpre = (torch.randn(C, C, device="cuda", generator=g) * C ** -0.5).to(torch.bfloat16), torch.randn(C, device="cuda", generator=g)
        wf5 = ops.tattn_pack(wqkv, wproj, wpre=wpre)
        mid = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)

        def five():                     # the spatial block's proj_out + residual in front
            ops.conv_gemm(att_s, wpre, bpre, residual=x, out=mid)
            ops.gn_small(mid, gamma, beta, geom, act=False, out=n1)
            ops.conv_gemm(n1, wqkv, bqkv, out=qkv)
            ops.attn_small(qkv, att, C, heads, geom)
            ops.conv_gemm(att, wproj, bproj, residual=mid, out=y0, stats=rec)

        def fused_pre():
            ops.tattn_block(x, wf5, bqkv, bproj, gamma, beta, heads, N, F, HW, out=y1, stats=rec, pre=(att_s, bpre, mid))

        four(); fused()
        err = float((y1.float() - y0.float()).norm() / y0.float().norm())
        t4 = tf = t5 = tp = 1e9
        for _ in range(3):
            t4 = min(t4, timed(four))
            tf = min(tf, timed(fused))
            t5 = min(t5, timed(five))
            tp = min(tp, timed(fused_pre))
        flops = 2.0 * M * C * 4 * C + 4.0 * M * F * C
        print(f"N={N} HW={HW:5d} C={C} M={M:6d} | four launches {t4:7.1f} us | fused {tf:7.1f} us  {flops / tf / 1e6:6.0f} TF/s  {3 * M * C * 2 / tf / 1e3:6.0f} GB/s (x + residual + y) | rel-L2 {err:.1e}"
              f" || with the spatial proj_out in front: five launches {t5:7.1f} us | fused {tp:7.1f} us")


if __name__ == "__main__":
    main()
