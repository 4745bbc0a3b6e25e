#!/usr/bin/env python3
"""One big GEMM shape, hot, for ablations of gemm_big.hip (CVFT_BIG_ABL)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
dev, dt = "cuda", torch.bfloat16
M, N, K = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (5328, 3072, 1024))]
R = int(sys.argv[4]) if len(sys.argv) > 4 else 16
PAD = int(os.environ.get("PAD", "0"))
x = torch.randn(M, K + PAD, device=dev, dtype=dt)[:, :K]
w = (torch.randn(N, K + PAD, device=dev, dtype=dt) / K ** 0.5)[:, :K]
o = torch.empty(M, N, device=dev, dtype=dt)
u, bl, b = torch.randn(M, R, device=dev, dtype=dt), torch.randn(N, R, device=dev, dtype=dt), torch.randn(N, device=dev)
for _ in range(3):
    HF.gemm(x, w, out=o, U=u, Bl=bl, bias=b)
ref = x.float() @ w.float().t() + b + u.float() @ bl.float().t()
err = float((o.float() - ref).norm() / ref.norm())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(50):
    HF.gemm(x, w, out=o, U=u, Bl=bl, bias=b)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 50 * 1e3
print(f"ABL={os.environ.get('CVFT_BIG_ABL', '0')} M{M} N{N} K{K}: {t:7.1f} us ({2.0 * M * N * (K + R) / t / 1e6:5.0f} TF/s) rel {err:.1e} [{HF.lib().cvft_gemm_last_kernel().decode()}]")
