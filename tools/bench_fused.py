#!/usr/bin/env python3
"""fused (in-launch U) vs two-launch LoRA linear forward, device time via hipGraph capture."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
from tools.bench_kernels import timeit
dev, dt = "cuda", torch.bfloat16
for M, N, K in [(4000, 512, 256), (8000, 512, 256), (4640, 512, 512), (4640, 2048, 512), (4640, 512, 2048), (5328, 1024, 1024),
                (5328, 4096, 1024), (5328, 1024, 4096)]:
    x = torch.randn(M, K, device=dev, dtype=dt)
    w = torch.randn(N, K, device=dev, dtype=dt) / K ** 0.5
    A = torch.randn(16, K, device=dev, dtype=dt) / K ** 0.5
    B = torch.randn(N, 16, device=dev, dtype=dt)
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=dt)
    U = torch.empty(M, 16, device=dev, dtype=dt)

    def two():
        u = HF.gemm(x, A, alpha=2.0, out=U)
        HF.gemm(x, w, bias=b, U=u, Bl=B, out=out)

    def fused():
        HF.gemm(x, w, bias=b, La=A, lora_scale=2.0, Uout=U, Bl=B, out=out)

    def plain():
        HF.gemm(x, w, bias=b, out=out)
    t2, tf, tp = timeit(two), timeit(fused), timeit(plain)
    print(f"M{M:5d} N{N:5d} K{K:5d}: two-launch {t2:7.1f} us   fused {tf:7.1f} us   no-LoRA {tp:7.1f} us")
