#!/usr/bin/env python3
"""GEMM timing with hot vs cold operands: NSETS distinct (x, w, out) buffer sets are cycled inside one graph, so with
many sets every launch reads memory that is cold in L2 / MALL / TLB (as in the training step)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF

dev, dt = "cuda", torch.bfloat16


MODE = sys.argv[1] if len(sys.argv) > 1 else "plain"      # plain | lora | fused


def call(x, w, o, u, bl, la):
    if MODE == "lora":
        HF.gemm(x, w, out=o, U=u, Bl=bl)
    elif MODE == "fused":
        HF.gemm(x, w, out=o, La=la, Bl=bl, lora_scale=2.0, Uout=u)
    else:
        HF.gemm(x, w, out=o)


def run(M, N, K, nsets, reps=120):
    sets = [(torch.randn(M, K, device=dev, dtype=dt), torch.randn(N, K, device=dev, dtype=dt), torch.empty(M, N, device=dev, dtype=dt),
             torch.randn(M, 16, device=dev, dtype=dt), torch.randn(N, 16, device=dev, dtype=dt), torch.randn(16, K, device=dev, dtype=dt))
            for _ in range(nsets)]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for st in sets[:2]:
            call(*st)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps):
            call(*sets[i % nsets])
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3


for M, N, K in [(4000, 256, 1024), (4000, 512, 256), (4000, 1024, 256), (16, 256, 1024), (5328, 1024, 1024)]:
    res = [run(M, N, K, ns) for ns in (1, 8, 60)]
    print(f"M{M} N{N} K{K}: hot {res[0]:6.1f} us   8 sets {res[1]:6.1f} us   60 sets {res[2]:6.1f} us")
