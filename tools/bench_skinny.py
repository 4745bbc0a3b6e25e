#!/usr/bin/env python3
"""Graph-timed rank-side products C[M,R] = X A^T at the step's shapes (cold operands cycled), checked against torch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
dev, dt = "cuda", torch.bfloat16
torch.manual_seed(0)
for M, R, K in [(4000, 48, 256), (4000, 48, 1536), (8000, 48, 256), (5328, 48, 1024), (5328, 48, 3072), (5328, 16, 1024), (5328, 16, 4096), (4640, 48, 512)]:
    nsets = 24
    sets = [(torch.randn(M, K, device=dev, dtype=dt), torch.randn(R, K, device=dev, dtype=dt) / K ** 0.5) for _ in range(nsets)]
    out = torch.empty(M, R, device=dev, dtype=dt)
    x, a = sets[0]
    HF.gemm(x, a, alpha=2.0, out=out)
    err = float((out.float() - 2.0 * x.float() @ a.float().t()).norm() / (2.0 * x.float() @ a.float().t()).norm())
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        HF.gemm(x, a, alpha=2.0, out=out)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for i in range(96):
            x, a = sets[i % nsets]
            HF.gemm(x, a, alpha=2.0, out=out)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / (3 * 96) * 1e3
    print(f"M{M} R{R} K{K}: {t:6.1f} us  {M * K * 2 / t / 1e3:7.0f} GB/s  rel {err:.1e}  [{HF.lib().cvft_gemm_last_kernel().decode()}]")
