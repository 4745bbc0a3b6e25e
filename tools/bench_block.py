#!/usr/bin/env python3
"""Estimator transformer block, second half (to_out + residual + norm3 + feed-forward): the row-tile chain kernels
(csrc/block_fused.hip) against the launch-per-stage form, forward and forward+backward, graph-timed (device time)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cosyvoice_lora_finetune_framework_amd import modules as Mo
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
from tools.bench_kernels import timeit

dev, dt = "cuda", torch.bfloat16


def main():
    torch.manual_seed(0)
    blk = Mo.BasicTransformerBlock(256, 8, 64, 0.0, "gelu").to(dev)
    for p in blk.parameters():
        p.requires_grad_(False)
    for M in (2000, 4000, 8000):
        o = torch.randn(M, 512, device=dev, dtype=dt)
        x0 = torch.randn(M, 256, device=dev, dtype=dt)
        dy = torch.randn(M, 256, device=dev, dtype=dt)
        for fuse in (False, True, "lean", "wide", "wide8"):
            HF.BLOCK_FUSE = bool(fuse)
            HF.BLOCK_LEAN = {"lean": "1", "wide": "2", "wide8": "4"}.get(fuse, "0")

            def fwd():
                with torch.no_grad():
                    blk._tail(o, x0, "gelu_erf")

            def fwdbwd():
                oo, xx = o.detach().requires_grad_(True), x0.detach().requires_grad_(True)
                blk._tail(oo, xx, "gelu_erf").backward(dy)
            tf, tb = timeit(fwd), timeit(fwdbwd)
            fl = 2.0 * M * 256 * (2 * 1024 + 512)
            print(f"M={M:5d} fused={fuse!s:5s}: fwd {tf:7.1f} us ({fl / tf / 1e6:6.1f} TF/s)   fwd+bwd {tb:7.1f} us  -> bwd ~{tb - tf:7.1f} us ({fl * 1.0 / max(tb - tf, 1e-3) / 1e6:6.1f} TF/s dgrad-only)")
    HF.BLOCK_FUSE = True


if __name__ == "__main__":
    main()
