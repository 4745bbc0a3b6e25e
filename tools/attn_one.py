#!/usr/bin/env python3
"""Run one attention shape fwd+bwd a few times (eager) for PMC profiling: attn_one.py B H L rel causal."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
B, H, L, rel, causal = (int(v) for v in sys.argv[1:6])
dev, dt = "cuda", torch.bfloat16
d = H * 64
q, k, v = (torch.randn(B * L, d, device=dev, dtype=dt, requires_grad=True) for _ in range(3))
ln = torch.full((B,), L, device=dev, dtype=torch.int32)
do = torch.randn(B * L, d, device=dev, dtype=dt)
if rel:
    p = torch.randn(2 * L - 1, d, device=dev, dtype=dt)
    bu, bv = torch.randn(H, 64, device=dev) * 0.1, torch.randn(H, 64, device=dev) * 0.1
for _ in range(3):
    o = HF.attn_relpos(q, k, v, p, bu, bv, B, H, L, ln, bool(causal), 0.125) if rel else HF.attn_bias(q, k, v, B, H, L, ln, 0.125)
    o.backward(do)
torch.cuda.synchronize()
