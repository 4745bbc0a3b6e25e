#!/usr/bin/env python3
"""Run one GEMM shape N times (for PMC profiling)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
M, N, K = (int(v) for v in sys.argv[1:4])
dt = torch.bfloat16
x = torch.randn(M, K, device="cuda", dtype=dt)
w = torch.randn(N, K, device="cuda", dtype=dt) / K ** 0.5
b = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=dt)
for _ in range(10):
    HF.gemm(x, w, bias=b, out=out)
torch.cuda.synchronize()
