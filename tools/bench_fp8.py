#!/usr/bin/env python3
"""fp8 GEMM (cvft_gemm_fp8) vs the bf16 kernel on the LLM shapes, rotating operand sets inside a hipGraph; also the
activation quantiser's cost.   usage: bench_fp8.py [M N K R]..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
dev, dt = "cuda", torch.bfloat16
shapes = [(5328, 4096, 1024, 16), (5328, 3072, 1024, 48), (5328, 1024, 4096, 16), (5328, 1024, 1024, 16), (9968, 4096, 1024, 64)]
nsets, reps = 8, 32


def timed(fn):
    fn(0)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(0)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps):
            fn(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


for M, N, K, R in shapes:
    sets = []
    for _ in range(nsets):
        x, w = torch.randn(M, K, device=dev, dtype=dt), torch.randn(N, K, device=dev, dtype=dt) / K ** 0.5
        u, bl, b = torch.randn(M, R, device=dev, dtype=dt), torch.randn(N, R, device=dev, dtype=dt), torch.randn(N, device=dev)
        xq, xs = HF.quant_fp8_rows(x)
        wq, ws = HF.quant_fp8_rows(w)
        sets.append((x, w, u, bl, b, xq, xs, wq, ws, torch.empty(M, N, device=dev, dtype=dt)))
    t_bf = timed(lambda i: HF.gemm(sets[i % nsets][0], sets[i % nsets][1], U=sets[i % nsets][2], Bl=sets[i % nsets][3], bias=sets[i % nsets][4], out=sets[i % nsets][9]))
    t_f8 = timed(lambda i: HF.gemm_fp8(sets[i % nsets][5], sets[i % nsets][6], sets[i % nsets][7], sets[i % nsets][8], U=sets[i % nsets][2], Bl=sets[i % nsets][3], bias=sets[i % nsets][4], out=sets[i % nsets][9]))
    t_q = timed(lambda i: HF.quant_fp8_rows(sets[i % nsets][0]))
    fl = 2.0 * M * N * (K + R)
    print(f"M{M} N{N} K{K} R{R}: bf16 {t_bf:6.1f} us ({fl / t_bf / 1e6:5.0f} TF/s)   fp8 {t_f8:6.1f} us ({fl / t_f8 / 1e6:5.0f} TF/s)   quant(x) {t_q:5.1f} us   fp8+quant {t_f8 + t_q:6.1f} us")
