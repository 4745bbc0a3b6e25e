#!/usr/bin/env python3
"""Graph-timed GEMM at the step's shapes for the tile configuration CVFT_GEMM_CFG / CVFT_GLDS_CFG selects
(0 = heuristics); every result is also checked against a torch fp32 product."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
from tools.bench_kernels import timeit

dev, dt = "cuda", torch.bfloat16
shapes = [(4000, 256, 512, 16), (4000, 512, 256, 16), (4000, 1024, 256, 0), (4000, 256, 1024, 0), (4000, 256, 256, 0),
          (8000, 1024, 256, 0), (8000, 256, 1024, 0), (8000, 512, 256, 16), (4640, 512, 512, 16), (4640, 2048, 512, 16),
          (4640, 512, 2048, 16), (5328, 1024, 1024, 16), (5328, 4096, 1024, 16), (5328, 1024, 4096, 16), (640, 1024, 1024, 16)]
res, worst = [], 0.0
torch.manual_seed(0)
for M, N, K, R in shapes:
    x = torch.randn(M, K, device=dev, dtype=dt)
    w = torch.randn(N, K, device=dev, dtype=dt) / K ** 0.5
    u = torch.randn(M, 16, device=dev, dtype=dt) if R else None
    bl = torch.randn(N, 16, device=dev, dtype=dt) if R else None
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=dt)
    HF.gemm(x, w, bias=b, U=u, Bl=bl, out=out)
    ref = x.float() @ w.float().t() + b
    if R:
        ref = ref + u.float() @ bl.float().t()
    err = float((out.float() - ref).norm() / ref.norm())
    worst = max(worst, err)
    t = timeit(lambda: HF.gemm(x, w, bias=b, U=u, Bl=bl, out=out))
    res.append(f"{t:7.1f}")
print(f"cfg {os.environ.get('CVFT_GEMM_CFG', '0'):>2s}/{os.environ.get('CVFT_GLDS_CFG', '0'):>2s}: " + " ".join(res) + f"  maxrel {worst:.1e}")
