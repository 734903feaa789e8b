#!/usr/bin/env python3
"""Per-block timeline of the direct-to-LDS conv GEMM (tile 129) for one shape of tools/gemm_bench.py.

Needs a debug build of the library (the product build has no instrumentation):
    MMD_EXTRA_CXXFLAGS=-DGEMM_TIMELINE python mm-diffusion_amd/build.py --force
    python tools/gemm_timeline.py <shape-index|all> [out.json]
    python mm-diffusion_amd/build.py --force          # back to the product build

Thread 0 of every block stamps the 100 MHz wall clock at: 0 block start, 1 first K step landed (prologue DMA + barrier),
2 main loop done, 3 stores issued; slot 4 = XCC / HW id.  Printed: kernel span, per-phase mean / p50 / p90 in us, the share
of a block's life each phase takes, blocks resident over time, and per-XCD block counts - i.e. where a short-K tile's
10 us go (fill latency vs loop vs epilogue) and whether the chip is evenly fed."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-diffusion_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mm_diffusion import _hip as H, ops  # noqa: E402
from gemm_bench import SHAPES  # noqa: E402

TICK_US = 0.01     # s_memrealtime: 100 MHz


def one(idx, tile=tile):
    name, M, Cin, taps, dims, Cout, res = SHAPES[idx]
    dt = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(M, Cin, device="cuda", generator=g).to(dt)
    w = (torch.randn(Cout, Cin * len(taps), device="cuda", generator=g) * (Cin * len(taps)) ** -0.5).to(dt)
    b = torch.randn(Cout, device="cuda", generator=g)
    r = torch.randn(M, Cout, device="cuda", generator=g).to(dt) if res else None
    y = torch.empty(M, Cout, device="cuda", dtype=dt)
    nblk = ((M + 127) // 128) * ((Cout + 127) // 128)
    buf = torch.zeros(nblk, 8, dtype=torch.int64, device="cuda")
    lib = H.lib()
    if not hasattr(lib, "mmd_debug_set_gemm_timeline"):
        raise SystemExit("library was not built with -DGEMM_TIMELINE (see the docstring)")
    lib.mmd_debug_set_gemm_timeline.argtypes = [ctypes.c_void_p]
    for _ in range(3):
        ops.conv_gemm(x, w, b, taps=taps, dims=dims, residual=r, out=y, tile=tile)
    torch.cuda.synchronize()
    assert lib.mmd_debug_set_gemm_timeline(ctypes.c_void_p(buf.data_ptr())) == 0
    ops.conv_gemm(x, w, b, taps=taps, dims=dims, residual=r, out=y, tile=tile)
    torch.cuda.synchronize()
    lib.mmd_debug_set_gemm_timeline(ctypes.c_void_p(0))
    t = buf.cpu().numpy().astype(np.int64)
    t0 = t[:, 0].min()
    start, fill, loop, epi = (t[:, 0] - t0) * TICK_US, (t[:, 1] - t[:, 0]) * TICK_US, (t[:, 2] - t[:, 1]) * TICK_US, (t[:, 3] - t[:, 2]) * TICK_US
    life = (t[:, 3] - t[:, 0]) * TICK_US
    span = (t[:, 3].max() - t0) * TICK_US
    xcc = (t[:, 4] >> 32) & 0xf

    def st(a):
        return f"mean {a.mean():6.2f}  p50 {np.percentile(a, 50):6.2f}  p90 {np.percentile(a, 90):6.2f}"
    print(f"== {name}  M={M} K={Cin * len(taps)} N={Cout}  blocks={nblk}  span {span:.1f} us  "
          f"(block-lives / span / 512 slots = {life.sum() / span / 512:.2f} occupancy of 2 blocks x 256 CUs)")
    print(f"   fill (launch -> first K step in LDS) {st(fill)}   {100 * fill.sum() / life.sum():4.1f} % of block life")
    print(f"   main loop                           {st(loop)}   {100 * loop.sum() / life.sum():4.1f} %")
    print(f"   epilogue (LDS transpose + stores)    {st(epi)}   {100 * epi.sum() / life.sum():4.1f} %")
    print(f"   block life                          {st(life)}")
    edges = np.linspace(0, span, 11)
    resident = [int(((start <= e) & (start + life > e)).sum()) for e in edges[:-1]]
    print("   blocks resident at 0,10,...,90 % of the span:", resident)
    print("   blocks per XCD:", np.bincount(xcc, minlength=8).tolist())
    return {"name": name, "span_us": float(span), "fill": float(fill.mean()), "loop": float(loop.mean()), "epilogue": float(epi.mean()),
            "life": float(life.mean()), "resident": resident}


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    tile = int(os.environ.get("TILE", "129"))          # TILE=130: the halo-tile kernel (eligible shapes only)
    out = [one(i, tile) for i in (range(len(SHAPES)) if which == "all" else [int(which)])]
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            json.dump(out, f, indent=1)
