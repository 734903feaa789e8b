#!/usr/bin/env python3
"""Do small launches of DIFFERENT HIP streams overlap on MI355X?  N launches of one kernel captured into one graph as 1, 2 and 4
independent chains (one stream each, forked from and joined to the origin): wall time per replay.  If the chip overlapped latency-bound
launches of different queues freely, 2 chains would take half the time of 1.  Shapes: the small-level launches of the Landscape model
at the lane's batch (2 samples).  usage: queue_overlap_probe.py [N]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mm-diffusion_amd"))
import torch  # noqa: E402
from mm_diffusion import ops  # noqa: E402


def cases():
    g = torch.Generator(device="cuda").manual_seed(0)

    def gemm(M, K, N, nbuf):
        xs = [torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16) for _ in range(nbuf)]
        w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
        b = torch.randn(N, device="cuda", generator=g)
        ys = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(nbuf)]
        return lambda i: ops.conv_gemm(xs[i], w, b, tile=131, out=ys[i])

    def apply(M, C, nbuf):
        xs = [torch.randn(M, C, device="cuda", generator=g).to(torch.bfloat16) for _ in range(nbuf)]
        a, b = torch.rand(2, C, device="cuda", generator=g) + 0.5, torch.randn(2, C, device="cuda", generator=g)
        geom = ops.Geom.per_sample(2, M // 2)
        ys = [torch.empty_like(x) for x in xs]
        return lambda i: ops.gn_apply(xs[i], a, b, geom, act=True, out=ys[i])

    yield "strip GEMM [2048,512,512] (ds8, 2 samples)", gemm(2048, 512, 512, 4)
    yield "strip GEMM [8192,384,384] (ds4, 2 samples)", gemm(8192, 384, 384, 4)
    yield "strip GEMM [32768,256,256] (ds2, 2 samples)", gemm(32768, 256, 256, 4)
    yield "strip GEMM [131072,128,128] (ds1, 2 samples)", gemm(131072, 128, 128, 4)
    yield "gn_apply [2048,512]", apply(2048, 512, 4)
    yield "gn_apply [32768,256]", apply(32768, 256, 4)


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    side = [torch.cuda.Stream() for _ in range(4)]
    keep = []          # graphs stay alive until the process ends (no graph / event destruction while anything may still replay)
    for name, fn in cases():
        for i in range(4):
            fn(i)
        torch.cuda.synchronize()
        line = f"{name:46s}"
        for chains in (1, 2, 4):
            gr = torch.cuda.CUDAGraph()
            cap = torch.cuda.Stream()
            with torch.cuda.stream(cap):
                gr.capture_begin()
                for c in range(chains):
                    side[c].wait_stream(cap)
                    with torch.cuda.stream(side[c]):
                        for _ in range(N // chains):
                            fn(c)
                for c in range(chains):
                    cap.wait_stream(side[c])
                gr.capture_end()
            best = 1e9
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                gr.replay()
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            line += f" | {chains} chain(s): {best * 1e6 / N:6.2f} us per launch"
            keep.append((gr, cap))
        print(line, flush=True)


if __name__ == "__main__":
    main()
