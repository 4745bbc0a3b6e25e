#!/usr/bin/env python3
"""The LLM chain's GEMM shapes (M = 16 x 333 rows) through cvft_gemm: correctness against an fp32 torch product on the
same operands, then graph-timed on rotating (cold) operand sets.  CVFT_P256=0 gives the 128x128 / 96x256 kernels for
comparison (run both in ONE gpurun call).   usage: bench_p256.py [check|time|both] [M]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
dev, dt = "cuda", torch.bfloat16
mode = sys.argv[1] if len(sys.argv) > 1 else "both"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 5328
SHAPES = [(3072, 1024, 48, "qkv fwd"), (1024, 1024, 16, "out fwd/dgrad"), (4096, 1024, 16, "w_1 fwd / w_2 dgrad"),
          (1024, 4096, 16, "w_2 fwd / w_1 dgrad"), (1024, 3072, 48, "qkv dgrad")]


def check():
    torch.manual_seed(0)
    for N, K, R, name in SHAPES:
        x = torch.randn(M, K, device=dev, dtype=dt)
        w = torch.randn(N, K, device=dev, dtype=dt) / K ** 0.5
        u = torch.randn(M, R, device=dev, dtype=dt)
        bl = torch.randn(N, R, device=dev, dtype=dt) * 0.1
        b = torch.randn(N, device=dev)
        res = torch.randn(M, N, device=dev, dtype=dt)
        pre = torch.empty(M, N, device=dev, dtype=dt)
        y = HF.gemm(x, w, U=u, Bl=bl, bias=b, act="relu", preact=pre, residual=res)
        label = HF.lib().cvft_gemm_last_kernel().decode()
        y2 = HF.gemm(x, w, U=u, Bl=bl, bias=b, act="relu", preact=torch.empty_like(pre), residual=res)
        z = x.float() @ w.float().t() + u.float() @ bl.float().t() + b
        ref = torch.relu(z) + res.float()
        e1 = float((pre.float() - z).norm() / z.norm())
        e2 = float((y.float() - ref).norm() / ref.norm())
        mx = float((y.float() - ref).abs().max())
        print(f"check M{M} N{N} K{K} R{R} [{label}]: preact rel {e1:.2e}  out rel {e2:.2e} max {mx:.3f}  rerun bitwise {bool(torch.equal(y, y2))}", flush=True)


def time_shape(N, K, R, name):
    nsets, reps = 8, 32
    sets = [(torch.randn(M, K, device=dev, dtype=dt), torch.randn(N, K, device=dev, dtype=dt) / K ** 0.5, torch.empty(M, N, device=dev, dtype=dt),
             torch.randn(M, max(R, 8), device=dev, dtype=dt), torch.randn(N, max(R, 8), device=dev, dtype=dt), torch.randn(N, device=dev)) for _ in range(nsets)]

    def call(i):
        x, w, o, u, bl, b = sets[i % nsets]
        HF.gemm(x, w, out=o, U=u if R else None, Bl=bl if R else None, bias=b)
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
    print(f"time  M{M} N{N} K{K} R{R} {name:22s}: {t:6.1f} us ({2.0 * M * N * (K + R) / t / 1e6:5.0f} TF/s) [{HF.lib().cvft_gemm_last_kernel().decode()}]", flush=True)


if len(sys.argv) > 3:          # custom shapes: N,K,R ...
    SHAPES = [tuple(int(v) for v in a.split(",")) + ("custom",) for a in sys.argv[3:]]
def stamps(N, K, R):
    """CVFT_P256_STAMP=1: s_memrealtime (100 MHz) stamps of wave 0 of every workgroup: start, prologue landed, loop done,
    extension done, epilogue issued, stores retired.  Launches back to back on rotating operands; reads the last one."""
    import ctypes as C, numpy as np
    sets = [(torch.randn(M, K, device=dev, dtype=dt), torch.randn(N, K, device=dev, dtype=dt) / K ** 0.5, torch.empty(M, N, device=dev, dtype=dt),
             torch.randn(M, max(R, 8), device=dev, dtype=dt), torch.randn(N, max(R, 8), device=dev, dtype=dt), torch.randn(N, device=dev)) for _ in range(6)]
    for i in range(12):
        x, w, o, u, bl, b = sets[i % 6]
        HF.gemm(x, w, out=o, U=u if R else None, Bl=bl if R else None, bias=b)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (512 * 8))()
    HF.lib().cvft_debug_p256_stamps(buf)
    a = np.array(buf, dtype=np.int64).reshape(512, 8)
    nwg = min(512, ((M + 255) // 256) * (N // 256))
    a = a[:nwg]
    t0 = a[:, 0].min()
    us = lambda v: v / 100.0
    print(f"stamps M{M} N{N} K{K} R{R} [{HF.lib().cvft_gemm_last_kernel().decode()}], {nwg} workgroups; us since the first workgroup's start:")
    for name, col in (("start", 0), ("prologue landed", 1), ("loop done", 2), ("extension done", 3), ("epilogue issued", 4), ("stores retired", 5)):
        v = us(a[:, col] - t0)
        print(f"   {name:16s} min {v.min():7.2f}  median {np.median(v):7.2f}  max {v.max():7.2f}")
    for name, c0, c1 in (("prologue", 0, 1), ("k-loop", 1, 2), ("extension", 2, 3), ("epilogue", 3, 4), ("store drain", 4, 5)):
        v = us(a[:, c1] - a[:, c0])
        print(f"   d {name:14s} min {v.min():7.2f}  median {np.median(v):7.2f}  max {v.max():7.2f}")


if mode == "stamps":
    for sh in (SHAPES if len(sys.argv) <= 3 else [tuple(int(v) for v in a.split(",")) + ("custom",) for a in sys.argv[3:]]):
        stamps(*sh[:3])
    sys.exit(0)
if mode in ("check", "both"):
    check()
if mode in ("time", "both"):
    for sh in SHAPES:
        time_shape(*sh)
