#!/usr/bin/env python3
"""Run the vendor-library GEMM on the LLM shapes once each (for `rocprofv3 --kernel-trace`: which macro-tiles it picks)."""
import torch
dev, dt = "cuda", torch.bfloat16
for M, N, K in [(5328, 4096, 1024), (5328, 1024, 4096), (5328, 3072, 1024), (5328, 1024, 3072), (5328, 1024, 1024), (4000, 768, 256), (4000, 256, 1024)]:
    x, w, b = torch.randn(M, K, device=dev, dtype=dt), torch.randn(N, K, device=dev, dtype=dt), torch.randn(N, device=dev, dtype=dt)
    for _ in range(3):
        torch.addmm(b, x, w.t())
    torch.cuda.synchronize()
