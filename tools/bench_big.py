#!/usr/bin/env python3
"""Large LLM-shape GEMMs, hot and cold, for tile experiments (CVFT_GLDS_BIG)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
dev, dt = "cuda", torch.bfloat16


def run(M, N, K, nsets, reps=48):
    sets = [(torch.randn(M, K, device=dev, dtype=dt), torch.randn(N, K, device=dev, dtype=dt) / K ** 0.5, torch.empty(M, N, device=dev, dtype=dt),
             torch.randn(M, 16, device=dev, dtype=dt), torch.randn(N, 16, device=dev, dtype=dt), torch.randn(N, device=dev)) for _ in range(nsets)]
    x, w, o, u, bl, b = sets[0]
    HF.gemm(x, w, out=o, U=u, Bl=bl, bias=b)
    ref = x.float() @ w.float().t() + b + u.float() @ bl.float().t()
    err = float((o.float() - ref).norm() / ref.norm())
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        HF.gemm(x, w, out=o, U=u, Bl=bl, bias=b)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps):
            x, w, o, u, bl, b = sets[i % nsets]
            HF.gemm(x, w, out=o, U=u, Bl=bl, bias=b)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3, err


for M, N, K in [(5328, 4096, 1024), (5328, 1024, 4096), (5328, 3072, 1024), (5328, 1024, 3072), (5328, 1024, 1024)]:
    th, err = run(M, N, K, 1)
    tc, _ = run(M, N, K, 12)
    fl = 2.0 * M * N * (K + 16)
    print(f"M{M} N{N} K{K}: hot {th:6.1f} us ({fl / th / 1e6:5.0f} TF/s)  cold {tc:6.1f} us ({fl / tc / 1e6:5.0f} TF/s)  rel {err:.1e} [{HF.lib().cvft_gemm_last_kernel().decode()}]")


def run_lib(M, N, K, nsets, reps=48):
    """Yardstick: the vendor library GEMM (torch.nn.functional.linear -> hipBLASLt) on the same operands (no LoRA term)."""
    import torch.nn.functional as F
    sets = [(torch.randn(M, K, device=dev, dtype=dt), torch.randn(N, K, device=dev, dtype=dt) / K ** 0.5, torch.randn(N, device=dev, dtype=dt))
            for _ in range(nsets)]
    outs = [torch.empty(M, N, device=dev, dtype=dt) for _ in range(nsets)]
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        torch.addmm(sets[0][2], sets[0][0], sets[0][1].t(), out=outs[0])
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps):
            x, w, b = sets[i % nsets]
            torch.addmm(b, x, w.t(), out=outs[i % nsets])
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3


if os.environ.get("CVFT_BENCH_LIB", "1") != "0":
    for M, N, K in [(5328, 4096, 1024), (5328, 1024, 4096), (5328, 3072, 1024), (5328, 1024, 3072), (5328, 1024, 1024), (4000, 768, 256), (4000, 1024, 256), (4000, 256, 1024)]:
        th, tc = run_lib(M, N, K, 1), run_lib(M, N, K, 12)
        fl = 2.0 * M * N * K
        print(f"hipBLASLt M{M} N{N} K{K}: hot {th:6.1f} us ({fl / th / 1e6:5.0f} TF/s)  cold {tc:6.1f} us ({fl / tc / 1e6:5.0f} TF/s)")
