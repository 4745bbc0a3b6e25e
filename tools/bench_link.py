#!/usr/bin/env python3
"""The estimator's block boundary: block i's tail followed by block i + 1's q|k|v head as two launches (cvft_block_tail_fwd, then
cvft_block_qkv_fwd in its 32-row / 64-row forms) against the linked launch (cvft_block_link_fwd), graph-timed (device time per
boundary; the 20 boundaries of one graph rotate over 4 weight sets, so the streams come from L2 / memory as in a step)."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from cosyvoice_lora_finetune_framework_amd.hipops import binding as cb
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
from cosyvoice_lora_finetune_framework_amd.hipops.blockpack import BlockLinkPack, BlockQkvPack, BlockTailPack
from test_block_fused_gpu import _qkv_case, _tail_args, _weights
from tools.bench_kernels import timeit

DEV = "cuda"


def main():
    p = float(os.environ.get("P", "0.05"))
    lib = cb.lib()
    HF._DROPOUT["seed"] = torch.full((1,), 1234, dtype=torch.int64, device=DEV)
    d = lambda t: t.to(DEV)
    for M in (2000, 4000):
        sets = []
        for k in range(4):
            w = {n: d(v) for n, v in _weights(512, 1024, seed=k).items()}
            tpack = BlockTailPack(w["wo"], w["bo"], w["gamma"], w["beta"], 1e-5, w["w1"], w["b1"], w["w2"], w["b2"])
            wq, _, _, _ = _qkv_case(64, p, seed=k + 9)
            hpack = BlockQkvPack(d(wq["wqkv"]), d(wq["bias"]), d(wq["gamma"]), d(wq["beta"]), 1e-5)
            A = d(torch.cat(wq["A"], 0).to(torch.bfloat16))
            Bb = torch.zeros(1536, 48)
            for t in range(3):
                Bb[512 * t:512 * (t + 1), 16 * t:16 * (t + 1)] = wq["B"][t]
            sets.append((tpack, hpack, BlockLinkPack(tpack, hpack), A, d(Bb.to(torch.bfloat16))))
        g = torch.Generator().manual_seed(M)
        o = d(torch.randn(M, 512, generator=g).to(torch.bfloat16))
        x0 = d(torch.randn(M, 256, generator=g).to(torch.bfloat16))
        bf = lambda *s: torch.empty(s, dtype=torch.bfloat16, device=DEV)
        x1, out, z = bf(M, 256), bf(M, 256), bf(-(-M // 64) * 64 * 1024)
        mean, rstd, mean2, rstd2 = (torch.empty(M, device=DEV) for _ in range(4))
        Y, U = bf(M, 1536), bf(M, 48)
        outs = [bf(M, 256) for _ in range(3)]

        def args(k, wide):
            tpack, hpack, lpack, A, Bb = sets[k]
            a = _tail_args(cb, M, o, x0, tpack, x1, out, mean, rstd, z, 3)
            q = cb.BlockQkvArgs()
            q.M, q.x, q.gamma, q.beta, q.eps, q.mean, q.rstd = M, cb.ptr(out), cb.ptr(hpack.gamma), cb.ptr(hpack.beta), 1e-5, cb.ptr(mean2), cb.ptr(rstd2)
            q.W_fwd, q.bias, q.N3, q.wide = cb.ptr(hpack.W_fwd), cb.ptr(hpack.bias), 1536, wide
            q.A, q.lda, q.Bb, q.ldb = cb.ptr(A), 256, cb.ptr(Bb), 48
            q.alpha, q.p = 2.0, p
            if p > 0:
                q.seed = cb.ptr(HF._DROPOUT["seed"])
                for i in range(3):
                    q.sites[i] = 5 + i
                    q.xd[i] = outs[i].data_ptr()
            else:
                q.y_out = cb.ptr(outs[0])
            q.U, q.ldu, q.Y, q.ldy = cb.ptr(U), 48, cb.ptr(Y), 1536
            return a, q, lpack
        state = {"k": 0}
        nsets = int(os.environ.get("SETS", "4"))      # SETS=1: one weight set, L2-warm streams (what a perfect prefetch would give)

        def pair(wide):
            a, q, _ = args(state["k"] % nsets, wide)
            state["k"] += 1
            cb.check(lib.cvft_block_tail_fwd(C.byref(a), cb.stream()), "tail")
            cb.check(lib.cvft_block_qkv_fwd(C.byref(q), cb.stream()), "head")

        def tail_only():
            a, q, _ = args(state["k"] % nsets, 0)
            state["k"] += 1
            cb.check(lib.cvft_block_tail_fwd(C.byref(a), cb.stream()), "tail")

        def linked():
            a, q, lpack = args(state["k"] % nsets, 0)
            state["k"] += 1
            cb.check(lib.cvft_block_link_fwd(C.byref(a), C.byref(q), cb.ptr(lpack.W_fwd), cb.stream()), "link")
        t0 = timeit(tail_only)
        t32, t64, tl = timeit(lambda: pair(0)), timeit(lambda: pair(1)), timeit(linked)
        print(f"M={M:5d} p={p}: fwd  tail alone {t0:6.1f} us | tail + head(32-row) {t32:6.1f} us | tail + head(64-row) {t64:6.1f} us | linked {tl:6.1f} us")
        # ---- backward: head backward (block i + 1) then tail backward (block i)
        linked()                                           # (statistics / z of the last weight set; values do not matter for timing)
        dY, dres = bf(M, 1536).normal_(), bf(M, 256).normal_()
        V, dx, dx1, do = bf(M, 48), bf(M, 256), bf(M, 256), bf(M, 512)

        def bargs(k, wide, lean):
            tpack, hpack, lpack, A, Bb = sets[k]
            At, Bbt = A.t().contiguous(), Bb.t().contiguous()
            keep.append((At, Bbt))
            b = cb.BlockQkvBwdArgs()
            b.M, b.dY, b.lddy, b.dres, b.x = M, cb.ptr(dY), 1536, cb.ptr(dres), cb.ptr(out)
            b.gamma, b.mean, b.rstd, b.N3, b.wide = cb.ptr(hpack.gamma), cb.ptr(mean2), cb.ptr(rstd2), 1536, wide
            b.W_bwd = cb.ptr(hpack.W_bwd_wide if wide else hpack.W_bwd)
            b.At, b.ldat, b.Bbt, b.ldbt = cb.ptr(At), 48, cb.ptr(Bbt), 1536
            b.alpha, b.p = 2.0, p
            if p > 0:
                b.seed = cb.ptr(HF._DROPOUT["seed"])
                for i in range(3):
                    b.sites[i] = 5 + i
            b.V, b.ldv, b.dx = cb.ptr(V), 48, cb.ptr(dx)
            t = cb.BlockTailBwdArgs()
            t.M, t.x1, t.dy, t.gamma, t.mean, t.rstd, t.z = M, cb.ptr(x1), cb.ptr(dx), cb.ptr(tpack.gamma), cb.ptr(mean), cb.ptr(rstd), cb.ptr(z)
            t.W_bwd = cb.ptr((tpack.W_bwd, tpack.W_bwd_lean, tpack.W_bwd_wide)[lean])
            t.F, t.DI, t.act, t.dx1, t.lean = 1024, 512, 3, cb.ptr(dx1), lean
            t.dout, t.lddo = cb.ptr(do), 512
            return b, t, lpack
        keep = []
        cache = {(k, w, l): bargs(k, w, l) for k in range(4) for (w, l) in ((0, 0), (1, 2))}

        def bpair(wide, lean):
            b, t, _ = cache[(state["k"] % nsets, wide, lean)]
            state["k"] += 1
            cb.check(lib.cvft_block_qkv_bwd(C.byref(b), cb.stream()), "head bwd")
            cb.check(lib.cvft_block_tail_bwd(C.byref(t), cb.stream()), "tail bwd")

        def blinked():
            b, t, lpack = cache[(state["k"] % nsets, 0, 0)]
            state["k"] += 1
            cb.check(lib.cvft_block_link_bwd(C.byref(b), C.byref(t), cb.ptr(lpack.W_bwd), cb.stream()), "link bwd")
        b32, b64, bl = timeit(lambda: bpair(0, 0)), timeit(lambda: bpair(1, 2)), timeit(blinked)
        print(f"M={M:5d} p={p}: bwd  head + tail (32-row) {b32:6.1f} us | head + tail (64-row) {b64:6.1f} us | linked {bl:6.1f} us")
    HF._DROPOUT["seed"] = None


if __name__ == "__main__":
    main()
