#!/usr/bin/env python3
"""The LoRA rank-side kernels at the step's shapes on rotating (cold) operand sets inside a hipGraph: forward
U = drop(x) A^T (+ the dropped copies), backward V = dY B, and the dA / dB slab pair.  Prints us and algorithmic GB/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
from cosyvoice_lora_finetune_framework_amd.hipops.functional import ptr, stream, check, lib, LoraGradSink

dev, dt = "cuda", torch.bfloat16
NSETS, REPS = 8, 32


def timeit(call):
    call(0)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        call(0)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(REPS):
            call(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * REPS) * 1e3


def main():
    HF.dropout_begin_step()
    ML = int(os.environ.get("LLM_ROWS", "2664"))      # 8 x 333: the LLM runs as two half-batch chains (5328 = whole batch)
    shapes = [("flow qkv (half batch)", 4000, 256, 48, 1536, 3), ("flow qkv (whole)", 8000, 256, 48, 1536, 3),
              ("LLM out", ML, 1024, 16, 1024, 1), ("LLM qkv", ML, 1024, 48, 3072, 3), ("LLM w_1", ML, 1024, 16, 4096, 1),
              ("LLM w_2", ML, 4096, 16, 1024, 1)]
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    for name, M, K, R, N, nsite in shapes:
        xs = [torch.randn(M, K, device=dev, dtype=dt) for _ in range(NSETS)]
        A = torch.randn(R, K, device=dev, dtype=dt) / K ** 0.5
        Bt = torch.randn(R, N, device=dev, dtype=dt) / N ** 0.5
        dYs = [torch.randn(M, N, device=dev, dtype=dt) for _ in range(NSETS)]
        Us = [torch.randn(M, R, device=dev, dtype=dt) for _ in range(NSETS)]
        sites = [HF._next_drop_site() for _ in range(nsite)]
        if "fwd" in only or not only:
            t = timeit(lambda i: HF.skinny_dropout(xs[i % NSETS], A, 2.0, 0.1, sites, keep_dropped=True))
            by = M * K * 2 * (1 + nsite) + M * R * 2
            print(f"{name:24s} U = drop(x) A^T        M{M} K{K} R{R}: {t:6.1f} us  {by / t / 1e3:6.0f} GB/s")
        if "v" in only or not only:
            t = timeit(lambda i: HF.gemm(dYs[i % NSETS], Bt, alpha=2.0))
            by = M * N * 2 + M * R * 2
            print(f"{name:24s} V = dY B               M{M} N{N} R{R}: {t:6.1f} us  {by / t / 1e3:6.0f} GB/s  [{lib().cvft_gemm_last_kernel().decode()}]")
        if "rank" in only or not only:
            rpa, nsa = LoraGradSink.plan(M, K)
            rpb, nsb = LoraGradSink.plan(M, N)
            wsA = torch.empty(nsa * R * K, device=dev, dtype=torch.float32)
            wsB = torch.empty(nsb * N * R, device=dev, dtype=torch.float32)
            def pair(i):
                x, V, dY, U = xs[i % NSETS], Us[(i + 1) % NSETS], dYs[i % NSETS], Us[i % NSETS]
                check(lib().cvft_lora_rank_partial_pair(M, R, K, ptr(x), x.stride(0), ptr(V), V.stride(0), ptr(wsA), rpa,
                                                        N, ptr(dY), dY.stride(0), ptr(U), U.stride(0), ptr(wsB), rpb, stream()), "pair")
            t = timeit(pair)
            by = M * K * 2 + M * N * 2 + 2 * M * R * 2 + (nsa * R * K + nsb * N * R) * 4
            print(f"{name:24s} dA, dB slabs ({nsa:2d}/{nsb:2d})   M{M} K{K} N{N} R{R}: {t:6.1f} us  {by / t / 1e3:6.0f} GB/s")


main()
