#!/usr/bin/env python3
"""Micro-benchmark of mmd_attn_fwd on the attention shapes of the Landscape model at batch 4 (bf16)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-diffusion_amd"))
import torch  # noqa: E402
from mm_diffusion import _hip as H, ops  # noqa: E402

N, F = 4, 16
SHAPES = [  # name, q rows/batch, q/group, k rows/batch, k/group, win, heads, ch
    ("spatial ds2 T=1024", F * 1024, 1024, F * 1024, 1024, 1, 4, 64),
    ("cross v<-a ds2", F * 1024, 1024, 6400, 400, 1, 4, 64),
    ("cross a<-v ds2", 6400, 400, F * 1024, 1024, 1, 4, 64),
    ("spatial ds4 T=256", F * 256, 256, F * 256, 256, 1, 4, 96),
    ("cross v<-a ds4 w4", F * 256, 256, 1600, 100, 4, 6, 64),
    ("cross a<-v ds4 w4", 1600, 100, F * 256, 256, 4, 6, 64),
]


def main():
    dt = torch.bfloat16
    ev = [ctypes.c_void_p(), ctypes.c_void_p()]
    for e in ev:
        H.call("mmd_event_create", ctypes.byref(e))
    st = H.stream_handle()
    only = os.environ.get("ATTN_BENCH_SHAPES")
    impls = tuple(int(v) for v in os.environ.get("ATTN_BENCH_IMPLS", "2,4,5").split(","))
    for si, (name, qr, qg, kr, kg, win, heads, ch) in enumerate(SHAPES):
        if only and str(si) not in only.split(","):
            continue
        C = heads * ch
        g = torch.Generator(device="cuda").manual_seed(0)
        q = torch.randn(N * qr, 3 * C, device="cuda", generator=g).to(dt)
        kv = torch.randn(N * kr, 3 * C, device="cuda", generator=g).to(dt)
        sh = torch.tensor([3], dtype=torch.int32, device="cuda")
        flops = 4.0 * N * qr * win * kg * C
        line = f"{name:22s} ({flops/1e9:5.1f} GF)"
        ref = None
        for impl in impls:             # 2 = per-128-query MFMA kernel, 3 = staged-window kernel, 4 = DMA-staged kernel, 5 = hand-written pipelined kernel (head width 64 only)
            if impl in (3, 4, 5) and ch != 64:
                continue
            out = torch.zeros(N * qr, C, device="cuda", dtype=dt)
            for _ in range(2):
                ops.attn(q, kv, out, heads, ch, N, F, qr, qg, kr, kg, win, shift_dev=sh, impl=impl)
            err = 0.0
            if ref is None:
                ref = out.clone()
            else:
                err = float((out.float() - ref.float()).norm() / ref.float().norm())
            # variant builds (MMD_LIB): ATTN_BENCH_SAVE=dir stores the outputs of this build, ATTN_BENCH_CMP=dir compares with them
            if os.environ.get("ATTN_BENCH_SAVE"):
                os.makedirs(os.environ["ATTN_BENCH_SAVE"], exist_ok=True)
                torch.save(out.cpu(), os.path.join(os.environ["ATTN_BENCH_SAVE"], f"s{si}_i{impl}.pt"))
            vs = ""
            if os.environ.get("ATTN_BENCH_CMP"):
                base = torch.load(os.path.join(os.environ["ATTN_BENCH_CMP"], f"s{si}_i{impl}.pt")).cuda().float()
                vs = f" vs-product {float((out.float() - base).norm() / base.norm()):.1e}"
            H.call("mmd_event_record", ev[0], st)
            n = 10
            for _ in range(n):
                ops.attn(q, kv, out, heads, ch, N, F, qr, qg, kr, kg, win, shift_dev=sh, impl=impl)
            H.call("mmd_event_record", ev[1], st)
            ms = ctypes.c_float()
            H.call("mmd_event_elapsed_ms", ev[0], ev[1], ctypes.byref(ms))
            us = ms.value / n * 1000
            line += f" | impl{impl}: {us:7.1f} us {flops/us/1e6:5.0f} TF/s ({100*flops/us/1e6/2500:4.1f}% mfma) e={err:.1e}{vs}"
        if ch == 64:                   # the DMA-staged kernel's exact-scale instance (training forward): the round-3 softmax loop, for A/B
            out = torch.zeros(N * qr, C, device="cuda", dtype=dt)
            lse = torch.zeros(N * qr, heads, device="cuda")
            for _ in range(2):
                ops.attn_lse(q, kv, out, lse, heads, ch, N, F, qr, qg, kr, kg, win, shift_dev=sh)
            H.call("mmd_event_record", ev[0], st)
            for _ in range(10):
                ops.attn_lse(q, kv, out, lse, heads, ch, N, F, qr, qg, kr, kg, win, shift_dev=sh)
            H.call("mmd_event_record", ev[1], st)
            ms = ctypes.c_float()
            H.call("mmd_event_elapsed_ms", ev[0], ev[1], ctypes.byref(ms))
            us = ms.value / 10 * 1000
            line += f" | dma-exact(+lse): {us:7.1f} us {flops/us/1e6:5.0f} TF/s"
        print(line, flush=True)


if __name__ == "__main__":
    main()
