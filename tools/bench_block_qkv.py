#!/usr/bin/env python3
"""Estimator transformer block, first half (norm1 + LoRA q|k|v, csrc/block_qkv.hip) and its backward through the C ABI: 32 rows per
workgroup against the 64-row form (csrc/block_qkv_wide.hip), graph-timed (device time)."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from cosyvoice_lora_finetune_framework_amd.hipops import binding as cb
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
from cosyvoice_lora_finetune_framework_amd.hipops.blockpack import BlockQkvPack
from test_block_fused_gpu import _qkv_case
from tools.bench_kernels import timeit

DEV = "cuda"


def main():
    p = float(os.environ.get("P", "0.05"))
    for M in (2000, 4000, 8000):
        w, x, dY, dres = _qkv_case(M, p, seed=M)
        d = lambda t: t.to(DEV)
        pack = BlockQkvPack(d(w["wqkv"]), d(w["bias"]), d(w["gamma"]), d(w["beta"]), 1e-5)
        A = torch.cat(w["A"], 0).to(torch.bfloat16)
        Bb = torch.zeros(1536, 48)
        for t in range(3):
            Bb[512 * t:512 * (t + 1), 16 * t:16 * (t + 1)] = w["B"][t]
        Bb = Bb.to(torch.bfloat16)
        ops = (d(A), d(A.t().contiguous()), d(Bb), d(Bb.t().contiguous()))
        HF._DROPOUT["seed"] = torch.full((1,), 1234, dtype=torch.int64, device=DEV)
        xd_ = d(x.to(torch.bfloat16))
        Y = torch.empty((M, 1536), dtype=torch.bfloat16, device=DEV)
        U = torch.empty((M, 48), dtype=torch.bfloat16, device=DEV)
        mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
        outs = [torch.empty((M, 256), dtype=torch.bfloat16, device=DEV) for _ in range(3)]
        V = torch.empty((M, 48), dtype=torch.bfloat16, device=DEV)
        dx = torch.empty((M, 256), dtype=torch.bfloat16, device=DEV)
        dYd, dresd = d(dY.to(torch.bfloat16)), d(dres.to(torch.bfloat16))
        for wide in (0, 1):
            a = cb.BlockQkvArgs()
            a.M, a.x, a.gamma, a.beta, a.eps, a.mean, a.rstd = M, cb.ptr(xd_), cb.ptr(pack.gamma), cb.ptr(pack.beta), 1e-5, cb.ptr(mean), cb.ptr(rstd)
            a.W_fwd, a.bias, a.N3, a.wide = cb.ptr(pack.W_fwd), cb.ptr(pack.bias), 1536, wide
            a.A, a.lda, a.Bb, a.ldb = cb.ptr(ops[0]), 256, cb.ptr(ops[2]), 48
            a.alpha, a.p = 2.0, p
            b = cb.BlockQkvBwdArgs()
            b.M, b.dY, b.lddy, b.dres, b.x = M, cb.ptr(dYd), 1536, cb.ptr(dresd), cb.ptr(xd_)
            b.gamma, b.mean, b.rstd, b.W_bwd, b.N3 = cb.ptr(pack.gamma), cb.ptr(mean), cb.ptr(rstd), cb.ptr(pack.W_bwd_wide if wide else pack.W_bwd), 1536
            b.wide = wide
            b.At, b.ldat, b.Bbt, b.ldbt = cb.ptr(ops[1]), 48, cb.ptr(ops[3]), 1536
            b.alpha, b.p = 2.0, p
            if p > 0:
                a.seed = b.seed = cb.ptr(HF._DROPOUT["seed"])
                for i in range(3):
                    a.sites[i] = b.sites[i] = 5 + i
                    a.xd[i] = outs[i].data_ptr()
            else:
                a.y_out = cb.ptr(outs[0])
            a.U, a.ldu, a.Y, a.ldy = cb.ptr(U), 48, cb.ptr(Y), 1536
            b.V, b.ldv, b.dx = cb.ptr(V), 48, cb.ptr(dx)
            lib = cb.lib()
            tf = timeit(lambda: cb.check(lib.cvft_block_qkv_fwd(C.byref(a), cb.stream()), "fwd"))
            tb = timeit(lambda: cb.check(lib.cvft_block_qkv_bwd(C.byref(b), cb.stream()), "bwd"))
            fl = 2.0 * M * 256 * 1536
            print(f"M={M:5d} p={p} wide={wide}: fwd {tf:6.1f} us ({fl / tf / 1e6:6.1f} TF/s)   bwd {tb:6.1f} us ({fl / tb / 1e6:6.1f} TF/s)")
    HF._DROPOUT["seed"] = None


if __name__ == "__main__":
    main()
