#!/usr/bin/env python3
"""Graph-timed fused attention forward and forward+backward at the step's shapes (bf16)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
from tools.bench_kernels import timeit

dev, dt = "cuda", torch.bfloat16
torch.manual_seed(0)
for name, B, H, L, rel, causal in [("estimator T250", 16, 8, 250, False, False), ("estimator T500", 16, 8, 500, False, False),
                                   ("flow enc  L290", 16, 8, 290, True, False), ("llm text  L40 ", 16, 16, 40, True, True),
                                   ("llm       L333", 16, 16, 333, True, True)]:
    d = H * 64
    q, k, v = (torch.randn(B * L, d, device=dev, dtype=dt, requires_grad=True) for _ in range(3))
    ln = torch.full((B,), L, device=dev, dtype=torch.int32)
    if rel:
        p = torch.randn(2 * L - 1, d, device=dev, dtype=dt)
        bu, bv = torch.randn(H, 64, device=dev) * 0.1, torch.randn(H, 64, device=dev) * 0.1
        f = lambda: HF.attn_relpos(q, k, v, p, bu, bv, B, H, L, ln, causal, 0.125)
    else:
        f = lambda: HF.attn_bias(q, k, v, B, H, L, ln, 0.125)
    do = torch.randn(B * L, d, device=dev, dtype=dt)

    def fb():
        o = f()
        o.backward(do)
        q.grad = k.grad = v.grad = None
    with torch.no_grad():
        tf = timeit(f)
    tfb = timeit(fb)
    fl = 4.0 * B * H * L * L * 64 * (0.5 if causal else 1.0)
    print(f"{name} B{B} H{H}: fwd {tf:7.1f} us ({fl / tf / 1e6:6.1f} TF/s)   fwd+bwd {tfb:7.1f} us  (bwd {tfb - tf:7.1f})")
