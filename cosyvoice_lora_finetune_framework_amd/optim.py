"""Flat-buffer AdamW + warmup-cosine LR schedule (reference train_joint.py:198-226, 349-360).

MI355X-first: all trainable (LoRA) parameters live in ONE contiguous fp32 buffer and their
gradients in another, so that (a) the data-parallel exchange is a single RCCL all-reduce of one
buffer, (b) grad-norm clipping is one reduction kernel and (c) the AdamW update is one fused
kernel whose learning rate / step / clip coefficient are read from device memory (no host
sync, hipGraph-replayable).  Arithmetic == torch.optim.AdamW(betas .9/.999, eps 1e-8) preceded
by torch.nn.utils.clip_grad_norm_ (checked in tests/test_ops_gpu.py::test_adamw_flat_matches_torch)."""
from __future__ import annotations

import math
import weakref
from typing import Iterable, List, Optional

import torch

from .hipops import binding as cb


def lr_lambda(step: int, warmup_steps: int, total_steps: int, min_lr: float, base_lr: float) -> float:
    """train_joint.py:211-216 (note the reference's pi = 3.14159 and fp32 cosine)."""
    if step < warmup_steps:
        return step / max(1, warmup_steps)
    progress = (step - warmup_steps) / max(1, total_steps - warmup_steps)
    return max(min_lr / base_lr, 0.5 * (1 + torch.cos(torch.tensor(progress * 3.14159)).item()))


class FlatAdamW:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 5e-5, weight_decay: float = 0.01,
                 betas=(0.9, 0.999), eps: float = 1e-8, max_grad_norm: float = 1.0):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        assert self.params, "no trainable parameters"
        dev = self.params[0].device
        assert dev.type == "cuda", "FlatAdamW runs on the HIP path only"
        n = sum(p.numel() for p in self.params)
        self.n = n
        self.flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            k = p.numel()
            assert p.dtype == torch.float32
            self.flat_p[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + k].view(p.shape)
            p.grad = self.flat_g[off:off + k].view(p.shape)
            off += k
        # compute-dtype shadows (bf16 copy + transposed bf16 copy) of every adapter, refreshed by ONE kernel per
        # optimiser step; LinearFn picks them up through P._cvft_shadow.  Stacked q|k|v operand sets (see
        # hipops.functional.QKVStack) add their own destination tiles to the same launch.
        self.flat_c = torch.empty(n, dtype=torch.bfloat16, device=dev)
        self.flat_t = torch.empty(n, dtype=torch.bfloat16, device=dev)
        self._offsets = {}
        self._tile_rows = []
        self._stacks = {}
        off = 0
        es = self.flat_c.element_size()
        for p in self.params:
            k = p.numel()
            rows = p.shape[0]
            cols = k // rows
            self._offsets[id(p)] = off
            p._cvft_opt = weakref.ref(self)
            if p.dim() == 2:
                p._cvft_shadow = (self.flat_c[off:off + k].view(rows, cols), self.flat_t[off:off + k].view(cols, rows))
            self._add_tiles(off, rows, cols, self.flat_c.data_ptr() + off * es, cols, self.flat_t.data_ptr() + off * es, rows)
            off += k
        self.tiles = None
        self.base_lr, self.wd, self.betas, self.eps, self.max_grad_norm = lr, weight_decay, betas, eps, max_grad_norm
        self.lr_dev = torch.full((1,), lr, dtype=torch.float32, device=dev)
        self.step_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.gnorm_part = torch.zeros(1024, dtype=torch.float32, device=dev)      # CVFT_SUMSQ_PARTS
        self.step_count = 0
        self.refresh_shadows()

    def _add_tiles(self, src_off: int, rows: int, cols: int, dst_ptr: int, dst_pitch: int, dstt_ptr: int, dstt_pitch: int):
        for tr in range(-(-rows // 32)):
            for tc in range(-(-cols // 32)):
                self._tile_rows.append((src_off, dst_ptr, dstt_ptr, rows | (cols << 32), tr | (tc << 32),
                                        dst_pitch | (dstt_pitch << 32)))
        self.tiles = None

    def stack_for(self, As, Bs):
        """Stacked / block-diagonal bf16 operands of three adapters that share their input (q|k|v):
        A [3r, K] and A^T [K, 3r];  B_blk [3N, 3r] (block i = B_i) and B_blk^T [3r, 3N].  Filled by the same
        per-step shadow launch as the per-parameter shadows.  Returns None when the parameters are not ours."""
        key = tuple(id(p) for p in (*As, *Bs))
        st = self._stacks.get(key)
        if st is not None:
            return st
        if any(id(p) not in self._offsets for p in (*As, *Bs)):
            return None
        r, K = As[0].shape
        N = Bs[0].shape[0]
        g = len(As)
        dev = self.flat_p.device
        A = torch.zeros((g * r, K), dtype=torch.bfloat16, device=dev)
        At = torch.zeros((K, g * r), dtype=torch.bfloat16, device=dev)
        Bb = torch.zeros((g * N, g * r), dtype=torch.bfloat16, device=dev)
        Bbt = torch.zeros((g * r, g * N), dtype=torch.bfloat16, device=dev)
        es = 2
        for i, (Ap, Bp) in enumerate(zip(As, Bs)):
            self._add_tiles(self._offsets[id(Ap)], r, K, A.data_ptr() + i * r * K * es, K, At.data_ptr() + i * r * es, g * r)
            self._add_tiles(self._offsets[id(Bp)], N, r, Bb.data_ptr() + (i * N * g * r + i * r) * es, g * r,
                            Bbt.data_ptr() + (i * r * g * N + i * N) * es, g * N)
        st = (A, At, Bb, Bbt)
        self._stacks[key] = st
        self.refresh_shadows()
        return st

    def refresh_shadows(self):
        if self.tiles is None:
            self.tiles = torch.tensor(self._tile_rows, dtype=torch.int64).to(self.flat_p.device).contiguous()
        cb.check(cb.lib().cvft_lora_shadow(self.tiles.shape[0], cb.ptr(self.tiles), cb.ptr(self.flat_p), cb.stream()),
                 "cvft_lora_shadow")
        for p in self.params:            # in-place edits of a master (load_lora_weights, ...) bump _version => on-the-fly cast
            p._cvft_shadow_ver = p._version

    def zero_grad(self):
        self.flat_g.zero_()

    def set_lr(self, lr: float):
        self.lr_dev.fill_(lr)

    def grad_norm(self, grad_scale: float = 1.0) -> torch.Tensor:
        """L2 norm of (grad_scale * grads) as a device scalar (reads the last step()'s reduction)."""
        return self.gnorm_sq.sqrt() * abs(grad_scale)

    def step(self, grad_scale: float = 1.0):
        """clip_grad_norm_(max_grad_norm) on grad_scale*g, then AdamW -- three kernels, no host sync."""
        self.step_count += 1
        self.step_dev.add_(1.0)
        L = cb.lib()
        cb.check(L.cvft_sumsq_ordered(self.n, cb.ptr(self.flat_g), cb.ptr(self.gnorm_part), cb.ptr(self.gnorm_sq), cb.stream()),
                 "cvft_sumsq_ordered")       # fixed summation order: the clip is bitwise equal on every DP replica
        cb.check(L.cvft_adamw_flat(self.n, cb.ptr(self.flat_p), cb.ptr(self.flat_g), cb.ptr(self.m), cb.ptr(self.v),
                                   cb.ptr(self.lr_dev), self.betas[0], self.betas[1], self.eps, self.wd,
                                   cb.ptr(self.step_dev), cb.ptr(self.gnorm_sq), float(self.max_grad_norm or 0.0),
                                   float(grad_scale), cb.stream()), "cvft_adamw_flat")
        self.refresh_shadows()

    def state_dict(self):
        return {"m": self.m.cpu(), "v": self.v.cpu(), "step": self.step_count, "flat_p": self.flat_p.cpu()}

    def load_state_dict(self, sd):
        self.m.copy_(sd["m"]); self.v.copy_(sd["v"]); self.flat_p.copy_(sd["flat_p"])
        self.step_count = int(sd["step"])
        self.step_dev.fill_(float(self.step_count))
        self.refresh_shadows()
