#!/usr/bin/env python3
"""One GEMM shape on rotating (cold) operand sets inside a hipGraph; prints us + kernel label.  For tile / stage
threshold experiments via the CVFT_GLDS_* environment hooks.   usage: bench_cfg.py M N K [R] [act]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
dev, dt = "cuda", torch.bfloat16
M, N, K = (int(v) for v in sys.argv[1:4])
R = int(sys.argv[4]) if len(sys.argv) > 4 else 0
act = sys.argv[5] if len(sys.argv) > 5 else None
nsets, reps = 12, 48
sets = [(torch.randn(M, K, device=dev, dtype=dt), torch.randn(N, K, device=dev, dtype=dt) / K ** 0.5, torch.empty(M, N, device=dev, dtype=dt),
         torch.randn(M, max(R, 8), device=dev, dtype=dt)[:, :R] if R else None, torch.randn(N, max(R, 8), device=dev, dtype=dt)[:, :R] if R else None,
         torch.randn(N, device=dev)) for _ in range(nsets)]
def call(i):
    x, w, o, u, bl, b = sets[i % nsets]
    HF.gemm(x, w, out=o, U=u, Bl=bl, bias=b, act=act)
call(0)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    call(0)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for i in range(reps):
        call(i)
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    g.replay()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / (5 * reps) * 1e3
print(f"M{M} N{N} K{K} R{R}: {t:6.1f} us ({2.0 * M * N * (K + R) / t / 1e6:5.0f} TF/s) [{HF.lib().cvft_gemm_last_kernel().decode()}] env={ {k: v for k, v in os.environ.items() if k.startswith('CVFT_')} }")
