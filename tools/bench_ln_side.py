#!/usr/bin/env python3
"""LayerNorm backward with the producer's mask copy (cvft_layernorm_bwd_mask) followed by the adapter's V = s * dxm B launch,
against the one launch that writes both (cvft_layernorm_bwd_mask_side), on rotating operand sets inside a hipGraph."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
from cosyvoice_lora_finetune_framework_amd.hipops.functional import ptr, stream, check, lib, dt

dev, bf = "cuda", torch.bfloat16
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
    for M in (2664, 5328):
        C = 1024
        xs = [torch.randn(M, C, device=dev, dtype=bf) for _ in range(NSETS)]
        dys = [torch.randn(M, C, device=dev, dtype=bf) for _ in range(NSETS)]
        drs = [torch.randn(M, C, device=dev, dtype=bf) for _ in range(NSETS)]
        dx, dxm = torch.empty(M, C, device=dev, dtype=bf), torch.empty(M, C, device=dev, dtype=bf)
        V = torch.empty(M, 16, device=dev, dtype=bf)
        gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        mean, rstd = torch.randn(M, device=dev) * 0.1, torch.rand(M, device=dev) + 0.5
        Bt = torch.randn(16, C, device=dev, dtype=bf) / 32
        seed = HF._DROPOUT["seed"]

        def plain(i):
            check(lib().cvft_layernorm_bwd_mask(dt(xs[0]), M, C, ptr(xs[i % NSETS]), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd),
                                                ptr(dys[i % NSETS]), ptr(drs[i % NSETS]), ptr(dx), 0.1, ptr(seed), 3, ptr(dxm), stream()), "m")

        def two(i):
            plain(i)
            HF.gemm(dxm, Bt, alpha=2.0)

        def one(i):
            check(lib().cvft_layernorm_bwd_mask_side(M, C, ptr(xs[i % NSETS]), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd),
                                                     ptr(dys[i % NSETS]), ptr(drs[i % NSETS]), ptr(dx), 0.1, ptr(seed), 3, ptr(dxm),
                                                     ptr(Bt), 16, 2.0, ptr(V), stream()), "s")
        print(f"M {M} C {C}: mask copy only {timeit(plain):6.1f} us   + V launch {timeit(two):6.1f} us   one launch {timeit(one):6.1f} us")


main()
