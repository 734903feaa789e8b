#!/usr/bin/env python3
"""What the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches on plain bf16 GEMMs of the conv layers' shapes (M x K x N, no taps,
no epilogue) on this GPU - the yardstick for the implicit-GEMM conv kernels of tools/gemm_bench.py (diagnostic, GPU)."""
import torch

SHAPES = [("3x3 ds1 128->128", 262144, 1152, 128), ("3x3 ds1 256->128", 262144, 2304, 128), ("k3t ds1 128->128", 262144, 384, 128),
          ("3x3 ds2 256->256", 65536, 2304, 256), ("k3t ds2 256->256", 65536, 768, 256), ("3x3 ds4 384->384", 16384, 3456, 384),
          ("3x3 ds8 512->512", 4096, 4608, 512), ("qkv ds2 256->768", 65536, 256, 768), ("proj ds2 256->256", 65536, 256, 256),
          ("1x1 ds1 128->128", 262144, 128, 128), ("big square 8192", 8192, 8192, 8192), ("big 65536x4096x4096", 65536, 4096, 4096)]
for name, M, K, N in SHAPES:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    for _ in range(3):
        y = a @ b.t()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    n = 10
    for _ in range(n):
        y = a @ b.t()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{name:24s} M={M:6d} K={K:5d} N={N:5d}  {us:8.1f} us  {2.0 * M * K * N / us / 1e6:7.0f} TFLOP/s", flush=True)
