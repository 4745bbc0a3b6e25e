#!/usr/bin/env python3
"""Micro-benchmarks of the GEMM-family kernels at the step's real shapes (HIP events, 50 reps)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF

dev = "cuda"
dt = torch.bfloat16


def timeit(fn, reps=20):
    """device time per call: the calls are captured into one hipGraph (no host launch overhead)."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # the clock ramps with load: replay for ~50 ms before the timed replays (a cold 3 ms burst reads 2-3x slow)
    nrep = int(os.environ.get("BENCH_REPLAYS", 40))
    for _ in range(nrep):
        g.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(nrep):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (nrep * reps) * 1e3   # us


def main():
    shapes = [(4000, 512, 256), (8000, 512, 256), (4000, 1536, 256), (4000, 1024, 256), (4000, 256, 1024), (4640, 512, 512),
              (4640, 2048, 512), (5328, 1024, 1024), (5328, 4096, 1024), (5328, 1024, 4096), (5328, 4097, 1024)]
    print("main GEMM (LoRA r=16 side path, bias):")
    for M, N, K in shapes:
        x = torch.randn(M, K, device=dev, dtype=dt)
        w = torch.randn(N, K, device=dev, dtype=dt) / K ** 0.5
        u = torch.randn(M, 16, device=dev, dtype=dt)
        bl = torch.randn(N, 16, device=dev, dtype=dt)
        b = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=dt)
        t = timeit(lambda: HF.gemm(x, w, bias=b, U=u, Bl=bl, out=out))
        t2 = timeit(lambda: torch.nn.functional.linear(x, w))
        fl = 2 * M * N * (K + 16)
        print(f"  M{M:5d} N{N:5d} K{K:5d}: {t:7.1f} us  {fl / t / 1e6:7.1f} TF/s   [{HF.lib().cvft_gemm_last_kernel().decode()}]   hipBLASLt {t2:7.1f} us {2*M*N*K/t2/1e6:7.1f} TF/s")
    if os.environ.get("ONLY_MAIN"):
        sys.exit(0)
    print("skinny (U = s x A^T):")
    for M, K in [(4000, 256), (8000, 256), (4640, 512), (5328, 1024), (5328, 4096)]:
        x = torch.randn(M, K, device=dev, dtype=dt)
        a = torch.randn(16, K, device=dev, dtype=dt)
        t = timeit(lambda: HF.gemm(x, a, alpha=2.0))
        print(f"  M{M:5d} K{K:5d}: {t:7.1f} us   {M * K * 2 / t / 1e3:7.1f} GB/s")
    print("rank accum (dA = V^T X):")
    for M, Cn in [(4000, 256), (8000, 256), (4000, 512), (4640, 512), (5328, 1024), (5328, 4096)]:
        x = torch.randn(M, Cn, device=dev, dtype=dt)
        v = torch.randn(M, 16, device=dev, dtype=dt)
        o = torch.zeros(16, Cn, device=dev)
        t = timeit(lambda: HF.rank_accum(x, v, o, False))
        print(f"  M{M:5d} C{Cn:5d}: {t:7.1f} us   {M * Cn * 2 / t / 1e3:7.1f} GB/s")



if __name__ == "__main__":
    main()
