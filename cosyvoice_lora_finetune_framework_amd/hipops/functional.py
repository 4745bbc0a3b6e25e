"""torch.autograd wrappers over the libcvft C ABI (include/cvft.h).

Every op here launches hand-written gfx950 kernels on torch's current stream; torch only
owns the memory.  Activations are 2-D, channel-last ``[rows, C]`` contiguous tensors
(rows = batch * time).  Backward passes produce input gradients and LoRA A/B gradients
only: every other parameter is frozen by ``lora.apply_lora_to_model``
(reference lora.py:214-216), so no frozen-weight / norm-affine gradients are computed.
"""
from __future__ import annotations

import contextlib
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import torch

from . import binding as cb
from .binding import ACT, GemmArgs, check, dt, lib, ptr, stream


PROFILE = None   # bench.py sets this to a list: every tap-GEMM launch is then bracketed by HIP events


class _Bracket:
    """bench.py's roofline leg (PROFILE is a list): HIP events around one attention launch on its launch stream, with the
    launch's algorithmic work -- FLOPs of the visible (query, key) pairs only, operands and results once."""

    def __init__(self, kernel: str, flop: float, nbytes: float):
        self.rec = {"kernel": kernel, "flop": flop, "bytes": nbytes} if PROFILE is not None else None

    def __enter__(self):
        if self.rec is not None:
            self.rec["start"] = torch.cuda.Event(enable_timing=True)
            self.rec["start"].record()
        return self

    def __exit__(self, *exc):
        if self.rec is not None and exc[0] is None:
            self.rec["end"] = torch.cuda.Event(enable_timing=True)
            self.rec["end"].record()
            PROFILE.append(self.rec)
        return False


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


def _rowc(t: torch.Tensor) -> torch.Tensor:
    """Row-contiguous 2-D view with a 16-byte aligned pitch is enough for the GEMM-family kernels (they take a
    leading dimension): column slices of a fused buffer (dq|dk|dv) are consumed in place, no copy."""
    if t.dim() == 2 and t.stride(1) == 1 and (t.stride(0) * t.element_size()) % 16 == 0 and t.data_ptr() % 16 == 0:
        return t
    return t.contiguous()


# ---------------------------------------------------------------------------------
# raw launches
# ---------------------------------------------------------------------------------
@dataclass
class Geo:
    """Row geometry of one tap-GEMM launch (see cvft_gemm in include/cvft.h)."""
    Tm: int
    Tin: int
    Tout: int
    in_stride: int = 1
    out_stride: int = 1
    out_off: int = 0
    taps: Tuple[int, ...] = (0,)


def gemm(x: torch.Tensor, W: torch.Tensor, *, N: Optional[int] = None, K: Optional[int] = None,
         bias: Optional[torch.Tensor] = None, U: Optional[torch.Tensor] = None, Bl: Optional[torch.Tensor] = None,
         alpha: float = 1.0, act: Optional[str] = None, preact: Optional[torch.Tensor] = None,
         dact_src: Optional[torch.Tensor] = None, dact: Optional[str] = None,
         residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
         geo: Optional[Geo] = None, nb: int = 1, in_len: Optional[torch.Tensor] = None,
         out_len: Optional[torch.Tensor] = None, out_rows: Optional[int] = None,
         La: Optional[torch.Tensor] = None, lora_scale: float = 1.0, Uout: Optional[torch.Tensor] = None,
         xdrop=None, odrop=None) -> torch.Tensor:
    """C = epi(alpha * (taps(x) @ W^T + U @ Bl^T) + bias); W is [N][ntaps*K] (k contiguous).
    odrop = (p, site): output dropout in the epilogue, C = residual + keep(site) / (1 - p) * epi(.) (cvft.h).
    xdrop = (p, sites): the U Bl^T term enters per 16-wide rank tile t as mask_t / (1 - p) * (U_t Bl_t^T), masks over the
    output elements (the lora_dropout dgrad; include/cvft.h)."""
    assert x.dim() == 2 and W.dim() == 2 and x.dtype == W.dtype
    N = W.shape[0] if N is None else N
    a = GemmArgs()
    a.dtype = dt(x)
    if geo is None:
        geo = Geo(Tm=x.shape[0], Tin=x.shape[0], Tout=x.shape[0])
        nb = 1
    ntaps = len(geo.taps)
    K = (W.shape[1] // ntaps) if K is None else K
    assert x.shape[1] >= K and W.shape[1] >= ntaps * K
    assert x.shape[0] == nb * geo.Tin, (x.shape, nb, geo)
    a.M, a.N, a.K = nb * geo.Tm, N, K
    a.Tm, a.Tin, a.Tout = geo.Tm, geo.Tin, geo.Tout
    a.in_stride, a.out_stride, a.out_off, a.ntaps = geo.in_stride, geo.out_stride, geo.out_off, ntaps
    for i in range(4):
        a.tap_off[i] = geo.taps[i] if i < ntaps else 0
    a.in_len, a.out_len = ptr(in_len), ptr(out_len)
    a.A, a.lda = ptr(x), x.stride(0)
    a.W, a.ldw = ptr(W), W.stride(0)
    if U is not None:
        assert Bl is not None and U.shape[0] == a.M and Bl.shape[0] == N and U.shape[1] == Bl.shape[1], (tuple(U.shape), a.M, N, None if Bl is None else tuple(Bl.shape))
        a.U, a.ldu, a.R = ptr(U), U.stride(0), U.shape[1]
        a.Bl, a.ldbl = ptr(Bl), Bl.stride(0)
    if La is not None:       # fused side path: U = lora_scale * x La^T computed inside the launch, written to Uout
        assert Bl is not None and U is None and La.shape[0] == Bl.shape[1] and Bl.shape[0] == N
        a.La, a.ldla, a.lora_scale, a.R = ptr(La), La.stride(0), float(lora_scale), La.shape[0]
        a.Bl, a.ldbl = ptr(Bl), Bl.stride(0)
        if Uout is not None:
            assert Uout.shape[0] == a.M and Uout.shape[1] >= a.R
            a.Uout, a.ldu = ptr(Uout), Uout.stride(0)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() >= N
    a.bias, a.alpha, a.act = ptr(bias), float(alpha), ACT[act]
    if out is None:
        rows = nb * geo.Tout if out_rows is None else out_rows
        out = torch.empty((rows, N), dtype=x.dtype, device=x.device)
    assert out.shape[0] == nb * geo.Tout and out.shape[1] >= N
    if preact is not None:
        a.preact, a.ldp = ptr(preact), preact.stride(0)
    if dact_src is not None:
        a.dact_src, a.ldd, a.dact = ptr(dact_src), dact_src.stride(0), ACT[dact]
    if residual is not None:
        assert residual.shape[0] == out.shape[0]
        a.residual, a.ldr = ptr(residual), residual.stride(0)
    a.C, a.ldc = ptr(out), out.stride(0)
    if xdrop is not None:
        assert U is not None and U.shape[1] % 16 == 0 and len(xdrop[1]) == U.shape[1] // 16
        a.xdrop_p, a.xdrop_seed = float(xdrop[0]), ptr(_DROPOUT["seed"])
        for i, st in enumerate(xdrop[1]):
            a.xdrop_sites[i] = st
    if odrop is not None:
        assert out.stride(0) == N and N % 4 == 0
        a.odrop_p, a.odrop_site, a.xdrop_seed = float(odrop[0]), int(odrop[1]), ptr(_DROPOUT["seed"])
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib().cvft_gemm(C.byref(a), stream()), "cvft_gemm")
        e1.record()
        r_eff = a.R if (U is not None or La is not None) else 0
        PROFILE.append({"kernel": lib().cvft_gemm_last_kernel().decode(),
                        "start": e0, "end": e1, "shape": (a.M, N, K, ntaps, a.R if (U is not None or La is not None) else 0),
                        "epi": "".join(c for c, on in (("b", bias is not None), ("a", act is not None), ("p", preact is not None),
                                                        ("d", dact_src is not None), ("r", residual is not None), ("o", odrop is not None),
                                                        ("x", xdrop is not None), ("f", La is not None)) if on),
                        "flop": 2.0 * a.M * N * (ntaps * K + r_eff) + (2.0 * a.M * K * a.R if La is not None else 0.0),
                        # algorithmic bytes: every operand and result once
                        "bytes": x.element_size() * (x.shape[0] * K + N * ntaps * K + a.M * N * (1 + (preact is not None) + (dact_src is not None) + (residual is not None))
                                                     + (a.M + N) * r_eff + (a.R * K if La is not None else 0))})
        return out
    check(lib().cvft_gemm(C.byref(a), stream()), "cvft_gemm")
    return out


def quant_fp8_rows(x: torch.Tensor):
    """(q uint8 [M, K] e4m3 bytes, scale fp32 [M]): q[m] = e4m3(x[m] / scale[m]), scale[m] = max|x[m]| / 448 (bf16 in)."""
    assert x.dim() == 2 and x.dtype == torch.bfloat16 and x.stride(1) == 1
    M, K = x.shape
    qt = torch.empty((M, K), dtype=torch.uint8, device=x.device)
    sc = torch.empty(M, dtype=torch.float32, device=x.device)
    check(lib().cvft_quant_fp8_rows(M, K, ptr(x), x.stride(0), ptr(qt), qt.stride(0), ptr(sc), stream()), "cvft_quant_fp8_rows")
    return qt, sc


def gemm_fp8(xq: torch.Tensor, xs: torch.Tensor, wq: torch.Tensor, ws: torch.Tensor, *, bias=None, U=None, Bl=None, alpha: float = 1.0,
             act: Optional[str] = None, preact=None, dact_src=None, dact: Optional[str] = None, residual=None, out=None,
             odrop=None) -> torch.Tensor:
    """C[M, N] bf16 = epi(alpha * (xs[m] ws[n] (xq wq^T) + U Bl^T) + bias) on e4m3 operands (quant_fp8_rows); LoRA term, bias,
    activation and residual as in gemm()."""
    M, K = xq.shape
    N = wq.shape[0]
    assert wq.shape[1] == K and xs.numel() == M and ws.numel() >= N
    a = GemmArgs()
    a.dtype = cb.BF16
    a.M, a.N, a.K = M, N, K
    a.Tm = a.Tin = a.Tout = M
    a.in_stride, a.out_stride, a.out_off, a.ntaps = 1, 1, 0, 1
    if U is not None:
        assert Bl is not None and U.shape[0] == M and Bl.shape[0] == N and U.shape[1] == Bl.shape[1]
        a.U, a.ldu, a.R = ptr(U), U.stride(0), U.shape[1]
        a.Bl, a.ldbl = ptr(Bl), Bl.stride(0)
    a.bias, a.alpha, a.act = ptr(bias), float(alpha), ACT[act]
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=xq.device)
    if preact is not None:
        a.preact, a.ldp = ptr(preact), preact.stride(0)
    if dact_src is not None:
        a.dact_src, a.ldd, a.dact = ptr(dact_src), dact_src.stride(0), ACT[dact]
    if residual is not None:
        a.residual, a.ldr = ptr(residual), residual.stride(0)
    a.C, a.ldc = ptr(out), out.stride(0)
    if odrop is not None:
        a.odrop_p, a.odrop_site, a.xdrop_seed = float(odrop[0]), int(odrop[1]), ptr(_DROPOUT["seed"])
    check(lib().cvft_gemm_fp8(C.byref(a), ptr(xq), xq.stride(0), ptr(xs), ptr(wq), wq.stride(0), ptr(ws), stream()), "cvft_gemm_fp8")
    return out


FP8_ON = False                 # BASELINE configs[4]: the frozen-W GEMMs of the big linears in e4m3 (opt-in: bench.py --fp8 1, CVFT_FP8=1)
FP8_MIN_WORK = 1 << 32         # M * N * K from which the activation quantisation pass pays (the LLM-sized linears)


def _mm(x: torch.Tensor, holder, key: str, W: torch.Tensor, **kw) -> torch.Tensor:
    """The frozen-weight GEMM of a linear (forward: holder.Wf, dgrad: holder.Wb) -- in e4m3 with per-row scales when FP8_ON and
    the shape is eligible (the weight's quantised copy is made once and cached on `holder`), else cvft_gemm as is."""
    N, K = W.shape
    if (FP8_ON and x.dtype == torch.bfloat16 and K % 128 == 0 and N % 4 == 0 and x.shape[0] * N * K >= FP8_MIN_WORK
            and x.shape[1] >= K and x.stride(1) == 1 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0 and W.is_contiguous()
            and not any(k in kw for k in ("La", "xdrop", "geo", "Uout"))):
        cache = holder.__dict__.setdefault("_fp8_w", {})
        ent = cache.get(key)
        if ent is None or ent[2] is not W:
            wq, ws = quant_fp8_rows(W)
            ent = cache[key] = (wq, ws, W)
        xq, xs = quant_fp8_rows(x[:, :K] if x.shape[1] != K else x)
        return gemm_fp8(xq, xs, ent[0], ent[1], **kw)
    return gemm(x, W, **kw)


def tn_accum(P: torch.Tensor, Q: torch.Tensor, G: torch.Tensor) -> None:
    """G[p,q] += sum_m P[m,p] Q[m,q]  (G fp32)."""
    assert P.shape[0] == Q.shape[0] and G.dtype == torch.float32 and G.shape == (P.shape[1], Q.shape[1])
    check(lib().cvft_tn_accum(dt(P), P.shape[0], P.shape[1], Q.shape[1], ptr(P), P.stride(0), ptr(Q), Q.stride(0),
                              ptr(G), G.stride(0), stream()), "cvft_tn_accum")


def act_fwd(x: torch.Tensor, act: str) -> torch.Tensor:
    x = _c(x)
    y = torch.empty_like(x)
    check(lib().cvft_act_fwd(dt(x), x.numel(), ACT[act], ptr(x), ptr(y), stream()), "cvft_act_fwd")
    return y


def act_bwd(z: torch.Tensor, dy: torch.Tensor, act: str) -> torch.Tensor:
    dz = torch.empty_like(z)
    check(lib().cvft_act_bwd(dt(z), z.numel(), ACT[act], ptr(z), ptr(_c(dy)), ptr(dz), stream()), "cvft_act_bwd")
    return dz


# ---------------------------------------------------------------------------------
# dropout (training mode of the encoders)
# ---------------------------------------------------------------------------------
_DROPOUT = {"seed": None, "site": 0}


_SEED_MASK = 0x7fffffffffff
_RANK_STRIDE = 0x9E3779B97F4A7C15      # per-rank offset of the mask stream: DP replicas must not draw the same masks


def _dist_rank() -> int:
    return torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0


def dropout_seed_state():
    """Rank-free state of the dropout stream (what a checkpoint stores): this rank's device seed minus its rank offset."""
    if _DROPOUT["seed"] is None:
        return None
    return (int(_DROPOUT["seed"].item()) - _RANK_STRIDE * _dist_rank()) & _SEED_MASK


def set_dropout_seed_state(base: int) -> None:
    """Resume the dropout stream from a checkpointed rank-free state: every rank re-applies ITS offset, and the value is copied
    INTO the existing seed tensor when there is one (a hipGraph captured earlier by this process increments that address)."""
    val = (int(base) + _RANK_STRIDE * _dist_rank()) & _SEED_MASK
    if _DROPOUT["seed"] is None:
        _DROPOUT["seed"] = torch.full((1,), val, dtype=torch.int64, device="cuda")
    else:
        _DROPOUT["seed"].fill_(val)


def dropout_begin_step() -> None:
    """Advance the device-side dropout seed (a captured op: every hipGraph replay draws new masks) and restart the
    call-site numbering.  JointLLMFlowModel.forward calls it once per training forward."""
    if _DROPOUT["seed"] is None:
        set_dropout_seed_state(int(torch.initial_seed()))
    _DROPOUT["seed"].add_(1)
    _DROPOUT["site"] = 0
    _DROPPED.clear()              # dropped inputs a previous forward wrote and no backward consumed (keyed by mask site)
    _PRE_DX.clear()               # parked fork gradients nobody took (keyed by a token drawn in forward)
    # (every other node-to-node hand-over rides ON the tensor object that is handed over -- _hand / _take_hand -- and dies with it)


class DropoutAddFn(torch.autograd.Function):
    """y = residual + dropout(x, p) (inverted dropout; residual optional); the mask is re-derived in backward."""

    @staticmethod
    def forward(ctx, x, residual, p: float, site: int):
        x = _c(x)
        y = torch.empty_like(x)
        res = None if residual is None else _c(residual)
        check(lib().cvft_dropout_add(dt(x), x.numel(), ptr(x), ptr(res), ptr(y), float(p), ptr(_DROPOUT["seed"]), site, stream()),
              "cvft_dropout_add")
        ctx.p, ctx.site, ctx.has_res = float(p), site, residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(dy)
            check(lib().cvft_dropout_add(dt(dy), dy.numel(), ptr(dy), None, ptr(dx), ctx.p, ptr(_DROPOUT["seed"]), ctx.site, stream()),
                  "cvft_dropout_add")
        return dx, (dy if ctx.has_res and ctx.needs_input_grad[1] else None), None, None


def dropout_add(x, p: float, residual=None):
    """residual + dropout(x, p); p == 0 degenerates to a plain add / identity."""
    if p <= 0.0:
        return x if residual is None else x + residual
    if _DROPOUT["seed"] is None:
        dropout_begin_step()
    _DROPOUT["site"] += 1
    return DropoutAddFn.apply(x, residual, p, _DROPOUT["site"])


class ActDropoutFn(torch.autograd.Function):
    """h = dropout(act(z), p): one pass forward, one pass backward (dz = keep/(1-p) * dh * act'(z)) -- instead of the
    activation in the GEMM epilogue + a dropout pass forward and a dropout pass + an activation-backward pass backward."""

    @staticmethod
    def forward(ctx, z, act: str, p: float, site: int):
        z = _c(z)
        h = torch.empty_like(z)
        check(lib().cvft_act_dropout(dt(z), z.numel(), ACT[act], ptr(z), None, ptr(h), float(p), ptr(_DROPOUT["seed"]), site, stream()),
              "cvft_act_dropout")
        ctx.save_for_backward(z)
        ctx.cfg = (act, float(p), site)
        return h

    @staticmethod
    def backward(ctx, dh):
        (z,) = ctx.saved_tensors
        act, p, site = ctx.cfg
        dh = _c(dh)
        dz = torch.empty_like(z)
        check(lib().cvft_act_dropout(dt(z), z.numel(), ACT[act], ptr(z), ptr(dh), ptr(dz), p, ptr(_DROPOUT["seed"]), site, stream()),
              "cvft_act_dropout")
        return dz, None, None, None


def act_dropout(z, act: str, p: float):
    """dropout(act(z), p) with p > 0 (training)."""
    return ActDropoutFn.apply(z, act, p, _next_drop_site())


def _next_drop_site() -> int:
    if _DROPOUT["seed"] is None:
        dropout_begin_step()
    _DROPOUT["site"] += 1
    return _DROPOUT["site"]


def _sites_arr(sites):
    return (C.c_uint * len(sites))(*sites)


def dropout_raw(x: torch.Tensor, p: float, site: int) -> torch.Tensor:
    """drop(x) for a given mask site (no autograd): re-materialises a forward mask in backward."""
    x = _c(x)
    y = torch.empty_like(x)
    check(lib().cvft_dropout_add(dt(x), x.numel(), ptr(x), None, ptr(y), float(p), ptr(_DROPOUT["seed"]), site, stream()),
          "cvft_dropout_add")
    return y


_DROPPED = {}       # mask site -> drop(x) written by the forward skinny kernel, consumed by that adapter's backward (dA = V^T drop(x))


# Node-to-node hand-overs.  A launch that makes a second product for the autograd node NEXT to it (the LayerNorm's rank-side U for
# the adapter that reads its output, a producer's mask site for the LayerNorm that reads ITS output, the masked gradient and side
# product a LayerNorm backward writes for the linear in front of it, the resnet fork's pairing) hands it over ON the tensor object
# that travels between the two nodes -- an attribute in the tensor's __dict__ -- never in a table keyed by the tensor's ADDRESS: an
# address can be recycled by the caching allocator between the two nodes (one wrong-gradient incident, round 3: a table keyed by
# x's address read in backward after x had been freed), an attribute cannot outlive or be detached from its tensor.  The receiver
# sees the same Python object whenever autograd passes the tensor through unchanged (one consumer, contiguous); when it does not
# (fan-in sum, a copy made by _c / a dtype cast), the attribute is simply absent and the receiver computes the product itself --
# every hand-over has that fallback, and every receiver still re-checks what it takes against its own identity ((p, site),
# (B^T address, scale), shapes).  Forward -> backward crossings use tokens drawn in forward (_DROPPED: mask site; _PRE_DX).
_H_PRE_U = "_cvft_pre_u"              # on y = LN(x):   (U, A data_ptr, alpha, p, sites, shape) from the LayerNorm launch
_H_ODROP = "_cvft_odrop"              # on y = residual + dropout(linear(.)):   (p, site, side) of the producing Function
_H_PRE_MASKED = "_cvft_pre_masked"    # on the dx a LayerNormForkFn backward returns:   (dxm, p, site, side product or None)
_H_PRE_V = "_cvft_pre_v"              # on dxm:   (V, Bt data_ptr, scale)
_H_FORK = "_cvft_fork"                # on x read by both convolutions of a ResnetBlock1D:   token of the "take" conv
_H_LINK = "_cvft_link"                # on the block output a linked tail launch wrote:   the NEXT block's head products (_QkvHead)
_H_LINK_BWD = "_cvft_link_bwd"        # on the dx a linked head backward returns:   the _LinkRec whose tail backward ran in that launch
_H_ATTN_O = "_cvft_attn_o"            # on the estimator attention's output o:   (o_lo or None, B, H, T) for the block tail that consumes o
_H_DELTA = "_cvft_delta"              # on the do a block tail backward returns:   delta [B, H, T] = rowsum(do . (o + o_lo)) per head


def _hand(t: torch.Tensor, name: str, value) -> None:
    t.__dict__[name] = value


HANDS_TAKEN = {}                      # name -> hand-overs a receiver found on its tensor (tests assert the path was exercised)


def _take_hand(t: torch.Tensor, name: str):
    v = t.__dict__.pop(name, None)
    if v is not None:
        HANDS_TAKEN[name] = HANDS_TAKEN.get(name, 0) + 1
    return v

# The residual-branch dropout of an encoder sublayer, y = residual + dropout(linear(.)), has its mask applied in the GEMM epilogue;
# in backward the linear needs keep / (1 - p) * dy.  dy is written by the LayerNormForkFn that consumed y (its one consumer), so
# that launch writes the masked copy too (cvft_layernorm_bwd_mask) and the linear's own mask pass over dy disappears.
# (_H_ODROP on y, noted by the producing Function and read by the LayerNormForkFn that takes y; _H_PRE_MASKED on the dx that
# LayerNormForkFn's backward returns)
# When that linear carries a rank-16 adapter, side = (Bt, scale) and the same LayerNorm-backward launch can also form
# V = scale * dxm Bt^T (cvft_layernorm_bwd_mask_side), the product the adapter's backward would open with as a launch of its own
# (40 launches per LLM step and chain): parked under dxm's address until _lin_bwd asks for exactly (dxm, Bt, scale).
# Opt-in (CVFT_LN_BWD_SIDE=1): per launch 9.0 / 16.5 us against 7.1 + 4.9 / 12.4 + 5.2 us for the two launches (2 664 / 5 328 rows,
# tools/bench_ln_side.py; matrix-core form: four rows per workgroup pass against B^T in LDS) -- less kernel time and one launch
# fewer, and still no faster in the step: same box, joint 21.48 / 21.54 (off) vs 21.55 / 21.57 (on), llm_only 13.49 vs 13.60 ms.
# (_H_PRE_V on dxm)
import os as _os  # noqa: E402
LN_BWD_MASK = _os.environ.get("CVFT_LN_BWD_MASK", "1") != "0"
LN_BWD_SIDE = _os.environ.get("CVFT_LN_BWD_SIDE", "0") != "0"


def _note_out_drop(y: torch.Tensor, od, side=None) -> None:
    """side = (Bt [16, N] compute dtype, scale) of the producing linear's adapter, or None."""
    if LN_BWD_MASK and od is not None:
        if not (LN_BWD_SIDE and side is not None and side[0] is not None and side[0].dtype == torch.bfloat16 and y.dtype == torch.bfloat16
                and side[0].dim() == 2 and side[0].shape[0] == 16 and side[0].shape[1] == y.shape[1] and side[0].is_contiguous()
                and side[0].data_ptr() % 16 == 0 and y.shape[1] % 128 == 0 and y.shape[1] <= 1536):
            side = None
        _hand(y, _H_ODROP, (float(od[0]), int(od[1]), side))


def _masked_dy(dy: torch.Tensor, od) -> torch.Tensor:
    """keep(site) / (1 - p) * dy: the copy the LayerNorm backward already wrote for exactly this (p, site), else one mask pass."""
    ent = _take_hand(dy, _H_PRE_MASKED)
    if ent is not None and ent[1] == float(od[0]) and ent[2] == int(od[1]) and ent[0].shape == dy.shape and ent[0].dtype == dy.dtype:
        if ent[3] is not None:
            _hand(ent[0], _H_PRE_V, ent[3])
        return ent[0]
    return dropout_raw(dy, od[0], od[1])


def _take_side_v(dz: torch.Tensor, Bt: torch.Tensor, scale: float):
    """The V = scale * dz Bt^T the LayerNorm backward that wrote dz made for exactly this (Bt, scale), else None."""
    ent = _take_hand(dz, _H_PRE_V)
    if ent is not None and ent[1] == Bt.data_ptr() and ent[2] == float(scale) and ent[0].shape == (dz.shape[0], Bt.shape[0]):
        return ent[0]
    return None


def _side_v(dz: torch.Tensor, Bt: torch.Tensor, scale: float) -> torch.Tensor:
    """V = scale * dz Bt^T [M, r]"""
    V = _take_side_v(dz, Bt, scale)
    return V if V is not None else gemm(dz, Bt, alpha=scale)


def take_pre_u(x: torch.Tensor, A: torch.Tensor, alpha: float, p: float, nsites: int):
    """(U, sites) that cvft_ln_skinny_dropout produced together with x = LN(.) for exactly this adapter, else None."""
    ent = _take_hand(x, _H_PRE_U)
    if ent is None:
        return None
    U, a_ptr, al, pp, sites, shape = ent
    if a_ptr == A.data_ptr() and al == float(alpha) and pp == float(p) and len(sites) == nsites and shape == tuple(x.shape):
        return U, sites
    for st in sites:                      # not the adapter it was made for: forget it and its dropped copies
        _DROPPED.pop(st, None)
    return None


def drop_pre_u(x: torch.Tensor) -> None:
    """forget a hand-off nobody took (the adapter went down another path after all)"""
    ent = _take_hand(x, _H_PRE_U)
    if ent is not None:
        for st in ent[4]:
            _DROPPED.pop(st, None)


def ln_skinny_dropout(x: torch.Tensor, gamma, beta, eps: float, A: torch.Tensor, alpha: float, p: float, nsites: int):
    """(y, mean, rstd) = LayerNorm(x) and, from the same launch, U = alpha * drop_t(y) A_t^T with the dropped copies kept for
    backward; U rides on y (_H_PRE_U) to the adapter Function that consumes y (take_pre_u)."""
    M, K = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x.device)
    U = torch.empty((M, A.shape[0]), dtype=x.dtype, device=x.device)
    sites = [_next_drop_site() for _ in range(nsites)]
    outs = [torch.empty_like(x) for _ in sites]
    for st, t in zip(sites, outs):
        _DROPPED[st] = t
    xd = (C.c_void_p * 3)(*([t.data_ptr() for t in outs] + [None] * (3 - len(outs))))
    check(lib().cvft_ln_skinny_dropout(M, K, A.shape[0], ptr(x), ptr(gamma), ptr(beta), float(eps), ptr(y), ptr(mean), ptr(rstd),
                                       ptr(A), A.stride(0), float(alpha), ptr(U), U.stride(0), float(p), ptr(_DROPOUT["seed"]),
                                       _sites_arr(sites), xd, stream()), "cvft_ln_skinny_dropout")
    _hand(y, _H_PRE_U, (U, A.data_ptr(), float(alpha), float(p), sites, (M, K)))
    return y, mean, rstd


def can_ln_skinny(x: torch.Tensor, A: torch.Tensor, gamma, beta) -> bool:
    return (LN_SKINNY and x.dtype == torch.bfloat16 and x.dim() == 2 and x.is_contiguous() and x.data_ptr() % 16 == 0
            and x.shape[1] % 32 == 0 and x.shape[1] <= 1024 and A.shape[0] in (16, 48) and A.stride(0) % 8 == 0
            and A.data_ptr() % 16 == 0 and gamma.data_ptr() % 16 == 0 and beta.data_ptr() % 16 == 0
            and not gamma.requires_grad and not beta.requires_grad)


def skinny_dropout(x: torch.Tensor, A: torch.Tensor, alpha: float, p: float, sites, keep_dropped: bool = False) -> torch.Tensor:
    """U[M, R] = alpha * drop_t(x) A_t^T per rank tile t (mask sites[t]); x bf16 contiguous [M, K], A [R, K].
    keep_dropped: the kernel also writes the dropped inputs (one per site) for the backward pass (dropped_input)."""
    U = torch.empty((x.shape[0], A.shape[0]), dtype=x.dtype, device=x.device)
    xd = None
    if len(sites) == 1 and A.shape[0] > 16:
        sites = list(sites) * (A.shape[0] // 16)          # ONE adapter of rank > 16: every rank tile under the same mask
    if keep_dropped and KEEP_DROPPED:
        uniq = list(dict.fromkeys(sites))                 # one dropped copy per distinct mask site
        outs = [torch.empty_like(x) for _ in uniq]
        for st, t in zip(uniq, outs):
            _DROPPED[st] = t
        xd = (C.c_void_p * 3)(*([t.data_ptr() for t in outs] + [None] * (3 - len(outs))))
    check(lib().cvft_skinny_dropout(x.shape[0], x.shape[1], A.shape[0], ptr(x), x.stride(0), ptr(A), A.stride(0), float(alpha),
                                    ptr(U), U.stride(0), float(p), ptr(_DROPOUT["seed"]), _sites_arr(sites), xd, stream()),
          "cvft_skinny_dropout")
    return U


def dropped_input(x: torch.Tensor, p: float, site: int) -> torch.Tensor:
    """drop_site(x): what the forward kernel wrote for this mask site, else re-derived from the counter-based mask."""
    t = _DROPPED.pop(site, None)
    return t if t is not None else dropout_raw(x, p, site)


def side_dgrad(V: torch.Tensor, A: torch.Tensor, dx: torch.Tensor, p: float, sites) -> torch.Tensor:
    """dx += sum_t mask_t/(1-p) * (V_t A_t)   in place (dx [M, K] bf16 contiguous rows)."""
    check(lib().cvft_lora_side_dgrad(dx.shape[0], dx.shape[1], A.shape[0], ptr(V), V.stride(0), ptr(A), A.stride(0), ptr(dx),
                                     dx.stride(0), ptr(dx), dx.stride(0), float(p), ptr(_DROPOUT["seed"]), _sites_arr(sites), stream()),
          "cvft_lora_side_dgrad")
    return dx


def _can_xdrop(dz: torch.Tensor, Wb: torch.Tensor, V: torch.Tensor, At: torch.Tensor, residual) -> bool:
    """dgrad with the masked rank extension inside the GEMM launch (gemm(..., xdrop=)): bf16, LDS-DMA register-epilogue
    kernels only -- K % 64 == 0, N % 4 == 0, N > 64, rank a multiple of 16 (<= 64), 16-byte aligned operands."""
    if not XDROP_ON or dz.dtype != torch.bfloat16:
        return False
    N, K, r = Wb.shape[0], Wb.shape[1], V.shape[1]
    ok = (K % 64 == 0 and N % 8 == 0 and N > 64 and r % 16 == 0 and r <= 64 and dz.stride(0) % 8 == 0 and Wb.stride(0) % 8 == 0
          and V.stride(0) % 8 == 0 and At.stride(0) % 8 == 0 and dz.data_ptr() % 16 == 0 and Wb.data_ptr() % 16 == 0
          and V.data_ptr() % 16 == 0 and At.data_ptr() % 16 == 0 and dz.shape[1] >= K and At.shape[0] == N)
    if residual is not None:
        ok = ok and residual.stride(0) % 4 == 0 and residual.data_ptr() % 8 == 0
    return ok


def _can_drop_fuse(x: torch.Tensor, r: int) -> bool:
    """The fused lora_dropout path (mask inside the skinny / side-dgrad kernels): bf16, contiguous rows, K % 32 == 0."""
    return (x.dtype == torch.bfloat16 and x.dim() == 2 and x.is_contiguous() and x.shape[1] % 32 == 0 and r in (16, 32, 48, 64)
            and x.data_ptr() % 16 == 0)


class ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act: str):
        x = _c(x)
        ctx.save_for_backward(x)
        ctx.act = act
        return act_fwd(x, act)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return act_bwd(x, dy, ctx.act), None


# ---------------------------------------------------------------------------------
# frozen-weight packs
# ---------------------------------------------------------------------------------
class LinearPack:
    """Frozen Linear weight in compute dtype: Wf [N][K] (forward), Wb [K][N] (dgrad), bias fp32."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], dtype: torch.dtype):
        self.N, self.K = weight.shape
        self.Wf = weight.detach().to(dtype).contiguous()
        self._Wb = None
        self.bias = None if bias is None else bias.detach().float().contiguous()
        # N not a multiple of the 16-byte vector (the LLM decoder: 4097 classes): zero-padded copies so that the
        # forward stores and the dgrad's K extent stay on the aligned kernels; see LinearFn
        vec = 16 // self.Wf.element_size()
        self.Npad = self.N if self.N % vec == 0 else -(-self.N // 64) * 64
        self._pad = None

    @property
    def padded(self):
        """(Wf_pad [Npad][K], bias_pad [Npad], Wb_pad [K][Npad]) with zero rows / columns past N."""
        if self._pad is None:
            Wf = self.Wf.new_zeros((self.Npad, self.K))
            Wf[:self.N] = self.Wf
            b = None
            if self.bias is not None:
                b = self.bias.new_zeros(self.Npad)
                b[:self.N] = self.bias
            self._pad = (Wf, b, Wf.t().contiguous())
        return self._pad

    @property
    def Wb(self) -> torch.Tensor:
        if self._Wb is None:
            self._Wb = self.Wf.t().contiguous()
        return self._Wb


def rank_accum(Wd: torch.Tensor, Rk: torch.Tensor, out: torch.Tensor, transpose_out: bool) -> None:
    """out[r,C] += Rk^T Wd (transpose_out=False) or out[C,r] += Wd^T Rk (True); out fp32, accumulated in place."""
    assert Wd.shape[0] == Rk.shape[0] and out.dtype == torch.float32 and out.is_contiguous()
    Cn, r = Wd.shape[1], Rk.shape[1]
    assert tuple(out.shape[:2]) == ((Cn, r) if transpose_out else (r, Cn))
    check(lib().cvft_lora_rank_accum(dt(Wd), Wd.shape[0], Cn, r, ptr(Wd), Wd.stride(0), ptr(Rk), Rk.stride(0), ptr(out),
                                     r if transpose_out else Cn, int(transpose_out), stream()), "cvft_lora_rank_accum")


class LoraGradSink:
    """Deterministic, atomic-free LoRA gradient accumulation for a whole backward pass:

        with LoraGradSink():
            loss.backward()

    Inside the context every LoRA layer writes per-row-block fp32 slabs of dA / dB into a persistent
    per-parameter workspace (cvft_lora_rank_partial); on exit ONE kernel (cvft_lora_grad_reduce) adds the
    slabs of all adapters into their .grad buffers in a fixed order.  Without an active sink the layers use
    the fp32-atomic kernel directly.  The task table lives on the device and is rebuilt only when the set of
    (workspace, grad, shape) tuples changes, so a captured hipGraph replays it unchanged."""
    active = None
    _cache = {}
    _side = None
    uses_hint = 1          # products expected per adapter and backward (JointLLMFlowModel: sub-batch chains of a branch)
    scattered = False      # a product got a slab buffer of its own since this was last cleared (train_joint._StepGraph warm-up)

    def __init__(self, side_stream: Optional[bool] = None):
        if side_stream is None:
            # measured: in a captured hipGraph every main<->side dependency edge crosses hardware queues (~9 us idle each,
            # ~4 ms / step); with the matrix-core slab kernels at ~3 us the serial order is faster
            side_stream = _os.environ.get('CVFT_SINK_SIDE', '0') != '0'
        self.tasks = []
        self.streams = {}                 # streams that received slab launches (the LLM / Flow branches may run on two)
        self.keep = []                    # operands of side-stream / deferred launches stay alive until the join
        self.deferred = {}                # rank -> table rows of slab products postponed to flush()
        self.uses = {}                    # id(parameter) -> [slab units handed out in this backward, their buffer]
        self.task_of = {}                 # (grad, geometry) -> index of its latest reduce task (merged when slabs are contiguous)
        self.side = None
        if side_stream:
            if LoraGradSink._side is None:
                LoraGradSink._side = torch.cuda.Stream()
            self.side = LoraGradSink._side

    def __enter__(self):
        assert LoraGradSink.active is None, "LoraGradSink is not re-entrant"
        LoraGradSink.active = self
        return self

    def __exit__(self, et, ev, tb):
        LoraGradSink.active = None
        if et is None:
            self.flush()
        return False

    @staticmethod
    def plan(M: int, Cn: int):
        """rows per slab block (64, 128 or k*256) and number of slabs for an [M, Cn] wide operand: the fewest
        slabs that still give >= 512 blocks (long row loops per block amortise the cross-wave reduction)."""
        colblocks = -(-Cn // 64)
        rpb = 64
        # slabs of a multiple of 128 rows run "stacked" (csrc/lora_grad.hip): four waves per 64-column stripe, each on a quarter
        # of the slab's rows -- four wave stripes per slab and column block
        for cand in (128, 256, 512, 768, 1024, 2048, 4096):
            if colblocks * SINK_STACK * (-(-M // cand)) >= SINK_PLAN_BLOCKS:
                rpb = cand
        return rpb, -(-M // rpb)

    def will_defer(self, x: torch.Tensor, dY: torch.Tensor) -> bool:
        """Small adapter-gradient products are postponed to flush() and issued there as a few chip-filling batch launches
        (cvft_lora_rank_partial_batch): nothing downstream in backward reads them, and one at a time they are
        launch-latency bound (~8 us each, ~250 per step on the flow branch's dgrad chain).  Large ones (the LLM's) stay
        in line: their operands are still in L2 / MALL there, and their branch is not the critical path."""
        return SINK_DEFER and self.side is None and x.dtype == torch.bfloat16 and x.numel() + dY.numel() <= SINK_DEFER_MAX

    @staticmethod
    def plan_deferred(M: int):
        """Batched launches fill the chip by their problem count: long row blocks (fewer, smaller slabs to write and reduce)."""
        return SINK_DEFER_RPB, -(-M // SINK_DEFER_RPB)

    def rank_pair(self, defer: bool, M: int, r: int, K: int, x, V, wsA, rpa: int, N: int, dY, U, wsB, rpb: int):
        """slabs of dA = V^T x ([r, K] per row block of rpa rows) and dB = dY^T U ([N, r] per row block of rpb rows): now, or
        at flush() when `defer`."""
        if not defer:
            check(lib().cvft_lora_rank_partial_pair(M, r, K, ptr(x), x.stride(0), ptr(V), V.stride(0), ptr(wsA), rpa,
                                                    N, ptr(dY), dY.stride(0), ptr(U), U.stride(0), ptr(wsB), rpb, stream()),
                  "cvft_lora_rank_partial_pair")
            return
        for C, Wd, Rk, ws, tr, rp in ((K, x, V, wsA, 0, rpa), (N, dY, U, wsB, 1, rpb)):
            assert (C % 8 == 0 and Wd.stride(0) % 8 == 0 and Rk.stride(0) % 8 == 0 and Wd.data_ptr() % 16 == 0 and Rk.data_ptr() % 16 == 0
                    and ws.data_ptr() % 16 == 0 and rp % 32 == 0 and Wd.dtype == Rk.dtype == torch.bfloat16 and Rk.shape[1] == r)
            self.deferred.setdefault(r, []).append((M, C, Wd.data_ptr(), Wd.stride(0), Rk.data_ptr(), Rk.stride(0), ws.data_ptr(), tr, rp,
                                                    torch.cuda.current_stream().cuda_stream))
        self.keep.append((x, V, dY, U))
        self._note_stream()
        self._deferred_added(r)

    def defer_one(self, M: int, r: int, C_: int, Wd, Rk, ws, transpose_out: int, rpb: int, keep=()):
        """Postpone ONE slab product (slabs of Rk^T Wd, [r, C] or [C, r] when transpose_out) to flush()."""
        assert (C_ % 8 == 0 and Wd.stride(0) % 8 == 0 and Rk.stride(0) % 8 == 0 and Wd.data_ptr() % 16 == 0 and Rk.data_ptr() % 16 == 0
                and ws.data_ptr() % 16 == 0 and rpb % 32 == 0 and Wd.dtype == Rk.dtype == torch.bfloat16 and Rk.shape[1] == r)
        self.deferred.setdefault(r, []).append((M, C_, Wd.data_ptr(), Wd.stride(0), Rk.data_ptr(), Rk.stride(0), ws.data_ptr(), transpose_out, rpb,
                                                torch.cuda.current_stream().cuda_stream))
        self.keep.append((Wd, Rk, ws) + tuple(keep))
        self._note_stream()
        self._deferred_added(r)

    @staticmethod
    def _launch_rows(r: int, rows):
        arr = (cb.RankProbM * len(rows))()
        for e, (M, Cn, Wd, ldw, Rk, ldr, ws, tr, rp, _) in zip(arr, rows):
            e.M, e.C, e.Wd, e.ldw, e.Rk, e.ldr, e.part, e.transpose_out, e.rows_per_block = M, Cn, Wd, ldw, Rk, ldr, ws, tr, rp
        check(lib().cvft_lora_rank_partial_batch(r, len(rows), arr, stream()), "cvft_lora_rank_partial_batch")

    def _launch_deferred(self):
        for r, rows in self.deferred.items():
            if rows:
                LoraGradSink._launch_rows(r, rows)
        self.deferred = {}

    def _deferred_added(self, r: int):
        """CVFT_SINK_DEFER_EARLY (default ON): once a chain has postponed SINK_EARLY_BATCH products of one rank, that batch goes out
        on the chain's OWN stream, mid-backward, instead of after the last chain has joined (the batch launches were ~0.9 ms of
        kernel time between the end of backward and the reduce).  A/B that justified the default, same box, joint B = 16:
        22.04 -> 21.67-21.73 ms for batches of 8-32 (flow_only 14.42 -> 14.32).  The rejected variant is a stream of their own for
        those batches: 22.2 -> 33.5 ms (a fifth stream in the captured graph shares a hardware queue with a chain).  Rows left over
        when a chain ends (fewer than a batch, or other chains' rows) leave with flush(); DESIGN section 13.  Covered by
        tests/test_ops_gpu.py::test_sink_early_batches_equal_end_of_backward_batches (adapter gradients EARLY=0 vs 1, two chains,
        a row count that is not a multiple of the batch)."""
        if not SINK_DEFER_EARLY:
            return
        cur = torch.cuda.current_stream()
        rows = self.deferred[r]
        mine = [i for i, e in enumerate(rows) if e[9] == cur.cuda_stream]
        if len(mine) < SINK_EARLY_BATCH:
            return
        LoraGradSink._launch_rows(r, [rows[i] for i in mine])
        keep = set(mine)
        self.deferred[r] = [e for i, e in enumerate(rows) if i not in keep]

    def flush_chain(self):
        """The postponed products of the CURRENT stream's chain go out now, on that stream (llm_flow_model.forward_backward calls
        this behind each chain's backward): what is left for flush() behind the join is the reduce alone."""
        cur = torch.cuda.current_stream().cuda_stream
        for r, rows in list(self.deferred.items()):
            mine = [e for e in rows if e[9] == cur]
            if mine:
                LoraGradSink._launch_rows(r, mine)
                self.deferred[r] = [e for e in rows if e[9] != cur]

    def workspace(self, P: torch.Tensor, nsplit: int) -> torch.Tensor:
        """Slab workspace (nsplit x P.numel() floats) for the next product on parameter P inside this sink.  Several
        products may hit the SAME parameter within one backward -- the sub-batch chains of a branch run the same adapters
        concurrently on their own streams (llm_flow_model.SPLIT); shared weights -- so every product gets its own slab
        range, laid out back to back in one per-parameter buffer: a chain never writes slabs another product owns, and
        the products' reduce tasks merge into ONE task per gradient (add / add_block; two tasks on one gradient in the
        same reduce launch would race on its read-modify-write)."""
        unit = P.numel()
        st = self.uses.get(id(P))
        if st is None:
            st = self.uses[id(P)] = [0, None]                 # units handed out in this backward, the buffer they come from
        total = st[0] + nsplit
        if st[1] is None:
            # what an earlier backward needed in all, or this product's share times the chains about to run (uses_hint)
            hint = max(getattr(P, "_cvft_ws_units", 0), total * max(1, LoraGradSink.uses_hint))
            buf = getattr(P, "_cvft_part", None)
            if buf is None or buf.numel() < hint * unit:
                # A captured step holds the ADDRESS of the workspace it was captured with; a later batch layout with more
                # row blocks needs a bigger one.  The old buffer is never freed (it stays that step's workspace): freeing
                # it let later allocations land under the slab writes of the older captured step (GPU memory fault in the
                # trainer's multi-layout path).  The new one gets headroom so that nearby layouts share it.
                if buf is not None:
                    P.__dict__.setdefault("_cvft_part_retired", []).append(buf)
                buf = torch.empty(hint * unit + hint * unit // 4, dtype=torch.float32, device=P.device)
                P._cvft_part = buf
            st[1] = buf
        if total * unit <= st[1].numel() and (unit % 4 == 0 or st[0] == 0):
            ws = st[1][st[0] * unit: total * unit]
        else:
            # more products on this parameter than any earlier backward had (or slabs that would lose their 16-byte
            # alignment): a buffer of its own now -- its reduce task goes into a later reduce launch (flush) -- and one
            # contiguous range from the next backward on
            ws = torch.empty(nsplit * unit, dtype=torch.float32, device=P.device)
            P.__dict__.setdefault("_cvft_part_retired", []).append(ws)
            LoraGradSink.scattered = True
        st[0] = total
        P._cvft_ws_units = max(getattr(P, "_cvft_ws_units", 0), total)
        return ws

    def _note_stream(self):
        st = torch.cuda.current_stream()
        self.streams[st.cuda_stream] = st

    def _task(self, part_ptr: int, grad: torch.Tensor, rows: int, cols: int, pitch: int, stride: int, nsplit: int):
        self._note_stream()
        key = (grad.data_ptr(), rows, cols, pitch, stride)
        i = self.task_of.get(key)
        if i is not None:
            t = self.tasks[i]
            if t[0] + t[6] * stride * 4 == part_ptr:          # the slabs follow the earlier product's: one longer task
                self.tasks[i] = t[:6] + (t[6] + nsplit, 0)
                return
        self.task_of[key] = len(self.tasks)
        self.tasks.append((part_ptr, grad.data_ptr(), rows, cols, pitch, stride, nsplit, 0))

    def add(self, part: torch.Tensor, grad: torch.Tensor, numel: int, nsplit: int):
        """grad (contiguous, numel) += sum of nsplit contiguous slabs."""
        self._task(part.data_ptr(), grad, 1, numel, numel, numel, nsplit)

    def add_block(self, part_ptr: int, grad: torch.Tensor, rows: int, cols: int, pitch: int, stride: int, nsplit: int):
        """grad [rows, cols] += sum over slabs of the [rows, cols] sub-block at part_ptr (row pitch / slab stride in floats)."""
        self._task(part_ptr, grad, rows, cols, pitch, stride, nsplit)

    def flush(self):
        if not self.tasks:
            return
        cur = torch.cuda.current_stream()
        if self.side is not None:
            cur.wait_stream(self.side)
        for h, st in self.streams.items():        # backward of a branch runs on the stream its forward used: join them
            if h != cur.cuda_stream:
                cur.wait_stream(st)
        self.streams = {}
        self._launch_deferred()
        self.keep = []
        key = tuple(self.tasks)
        ent = LoraGradSink._cache.get(key)
        if ent is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("LoraGradSink.flush: reduce task table not built before capture (run the step once more "
                                   "eagerly: the warm-up's slab layout differed from this one)")
            dev = torch.device("cuda", torch.cuda.current_device())
            # one reduce launch adds every task's slabs into its gradient with a plain read-modify-write: tasks that
            # share a gradient (products whose slabs could not be merged) go into successive launches
            waves, seen = [], []
            for t in self.tasks:
                w = next((i for i, g in enumerate(seen) if t[1] not in g), None)
                if w is None:
                    waves.append([])
                    seen.append(set())
                    w = len(waves) - 1
                waves[w].append(t)
                seen[w].add(t[1])
            ent = [(torch.tensor(w, dtype=torch.int64).to(dev), len(w), min(64, max(1, -(-max(t[2] * t[3] for t in w) // 1024))))
                   for w in waves]
            # (entries are never dropped: a captured step holds the address of its task tables)
            LoraGradSink._cache[key] = ent
        for tbl, n, gx in ent:
            check(lib().cvft_lora_grad_reduce(n, ptr(tbl), gx, stream()), "cvft_lora_grad_reduce")
        self.tasks = []
        self.task_of = {}
        self.uses = {}


import os as _os
if _os.environ.get('CVFT_FP8', '0') == '1':
    FP8_ON = True
# the stacked q|k|v adapters' four slab products join the sink's end-of-backward batch launches: 26.27 -> 25.91 ms/step with three
# chains (same-box A/B; it lost 0.7 ms when the step was one chain)
STACKED_DROP_DEFER = _os.environ.get('CVFT_STACKED_DROP_DEFER', '1') != '0'
KEEP_DROPPED = _os.environ.get('CVFT_KEEP_DROPPED', '1') != '0'   # forward writes drop(x) for the backward's dA (no re-derivation launch)
LN_SKINNY = _os.environ.get('CVFT_LN_SKINNY', '1') != '0'    # LayerNorm launch also emits the dropped rank-side product of the adapter it feeds
XDROP_ON = _os.environ.get('CVFT_XDROP', '1') != '0'        # lora_dropout dgrad: masked rank extension inside the GEMM launch
SINK_PLAN_BLOCKS = int(_os.environ.get('CVFT_SINK_PLAN_BLOCKS', 768))     # wave stripes per launch; with the stacked slab kernel (four waves per stripe and slab), same box: 512 -> 21.57, 768 -> 21.45 / 21.45, 1024 -> 21.56, 1536 -> 21.61 ms (the kernel before it: 21.70; round 3 had 512 best of 256 / 512 / 1024 for one wave per stripe, round 2 256)
SINK_STACK = int(_os.environ.get('CVFT_SINK_STACK', 4))       # wave stripes the planner counts per slab and column block (1: plan as before the stacked kernel)
SINK_DEFER = _os.environ.get('CVFT_SINK_DEFER', '1') != '0'
SINK_DEFER_MAX = int(_os.environ.get('CVFT_SINK_DEFER_MAX', 12_000_000))      # x.numel() + dY.numel(): the flow branch's layers
SINK_DEFER_RPB = int(_os.environ.get('CVFT_SINK_DEFER_RPB', 1024))     # stacked: 256 rows per wave as before, a quarter of the slabs (256 / 512 / 1024 / 2048: 22.04 / 21.81 / 21.80 / 21.20 vs 21.14 on a second box)
SINK_DEFER_EARLY = _os.environ.get('CVFT_SINK_DEFER_EARLY', '1') != '0'     # full batches leave during backward on their chain's stream
SINK_EARLY_BATCH = int(_os.environ.get('CVFT_SINK_EARLY_BATCH', 24))       # (= CVFT_RANK_BATCH, one launch)
FUSE_MAX_MN = int(_os.environ.get('CVFT_FUSE_MAX_MN', 3_000_000))   # measured (tools/bench_fused.py): the in-launch side path wins for the small estimator GEMMs only


def _can_fuse(x: torch.Tensor, La: torch.Tensor, Bl: torch.Tensor, N: int, K: int) -> bool:
    """Conditions of the in-launch side path (cvft_gemm `La`): rank <= 16, N > 32, 16-byte aligned operands,
    and a problem small enough that saving a launch beats the extra MFMAs of the fused tile."""
    vec = 8 if x.dtype == torch.bfloat16 else 4
    return (La.shape[0] <= 16 and N > 32 and x.shape[0] * N <= FUSE_MAX_MN and K % vec == 0 and x.stride(0) % vec == 0 and La.stride(0) % vec == 0
            and Bl.stride(0) % vec == 0 and x.data_ptr() % 16 == 0 and La.data_ptr() % 16 == 0 and Bl.data_ptr() % 16 == 0)


def _lora_operands(P: torch.Tensor, dtype):
    """(compute-dtype copy, transposed copy) of a LoRA master; uses the per-step shadows maintained by
    optim.FlatAdamW when present (one kernel per step for all adapters), else casts on the fly."""
    sh = getattr(P, "_cvft_shadow", None)
    if sh is not None and dtype == torch.bfloat16 and getattr(P, "_cvft_shadow_ver", -1) == P._version:
        return sh
    Pc = P.detach().to(dtype)
    return _c(Pc), Pc.t().contiguous()


# zero-padded row-pitch buffers handed between ops: data_ptr -> (weakref to the [M][pitch] base, pitch).  A [:, :N]
# view found here may be widened back to the base (its pad columns are zero) instead of being copied.
_ZERO_PADDED = {}


def _register_zero_padded(base: torch.Tensor) -> None:
    import weakref
    if len(_ZERO_PADDED) > 64:
        for k in [k for k, (w, _) in _ZERO_PADDED.items() if w() is None]:
            del _ZERO_PADDED[k]
    _ZERO_PADDED[base.data_ptr()] = (weakref.ref(base), base.shape[1])


def _widen_zero_padded(t: torch.Tensor, pitch: int) -> torch.Tensor:
    """[M][N] -> [M][pitch] with zero pad columns (no copy when `t` is a registered zero-padded view)."""
    ent = _ZERO_PADDED.get(t.data_ptr())
    if ent is not None and ent[0]() is not None and ent[1] == pitch and t.stride() == (pitch, 1) and \
            ent[0]().shape[0] == t.shape[0] and ent[0]().dtype == t.dtype:
        return ent[0]()
    w = t.new_zeros((t.shape[0], pitch))
    w[:, :t.shape[1]] = t
    return w


def _lin_fwd(x, A, B, pack: LinearPack, scale: float, act: Optional[str], residual, keep_preact: bool, drop=None, odrop=None):
    """One LoRA linear forward: y = act(x W^T + b + scale * (x A^T) B^T) (+ residual).
    Returns (y, U, z, ops): U = scale * x A^T [M, r] (saved for dB), z = pre-activation (when kept), ops = the
    compute-dtype (A, A^T, B, B^T) operands."""
    U = z = ops = None
    fused = False
    if A is not None:
        Ac, At = _lora_operands(A, x.dtype)
        Bc, Bt = _lora_operands(B, x.dtype)
        ops = (Ac, At, Bc, Bt)
        fused = drop is None and _can_fuse(x, Ac, Bc, pack.N, pack.K)
        if drop is not None:              # lora_dropout: the side path sees drop(x); mask applied inside the skinny kernel
            pre = take_pre_u(x, Ac, scale, drop[0], 1)        # made by the LayerNorm launch that produced x?
            if pre is not None:
                U = pre[0]
                drop[1] = pre[1][0]                           # its mask site replaces the one the caller drew
            else:
                U = skinny_dropout(x, Ac, scale, drop[0], [drop[1]], keep_dropped=True)
        elif fused:
            U = torch.empty((x.shape[0], Ac.shape[0]), dtype=x.dtype, device=x.device)
        else:
            U = gemm(x, Ac, alpha=scale)
    if act and keep_preact:
        z = torch.empty((x.shape[0], pack.N), dtype=x.dtype, device=x.device)
    res = None if residual is None else _c(residual)
    if fused:
        y = gemm(x, pack.Wf, bias=pack.bias, La=ops[0], lora_scale=scale, Uout=U, Bl=ops[2], act=act, preact=z, residual=res,
                 odrop=odrop)
    else:
        y = _mm(x, pack, 'f', pack.Wf, bias=pack.bias, U=U, Bl=None if ops is None else ops[2], act=act, preact=z, residual=res,
                odrop=odrop)
    return y, U, z, ops


def _lora_param_grads(x, U, V, dz, A_ref, B_ref, ops):
    """dA[r,K] += V^T x,  dB[N,r] += dz^T U  (V = s dz B, U = s x A^T): into the parameters' flat .grad buffers -- through the
    active LoraGradSink (deterministic slabs + one reduce) when there is one, else fp32 atomics -- or returned."""
    dA = dB = None
    A, B = A_ref, B_ref
    gA, gB = A.grad, B.grad
    direct = gA is not None and gB is not None and gA.dtype == torch.float32 and gA.is_contiguous() \
        and gB.is_contiguous() and gA.dim() == 2 and gB.dim() == 2
    if not direct:
        gA = torch.zeros(ops[0].shape, dtype=torch.float32, device=x.device)
        gB = torch.zeros(ops[2].shape, dtype=torch.float32, device=x.device)
    sink = LoraGradSink.active
    r = V.shape[1]
    vec = 8 if x.dtype == torch.bfloat16 else 4
    if (sink is not None and direct and r % 16 == 0 and x.shape[1] % vec == 0 and dz.shape[1] % vec == 0
            and x.data_ptr() % 16 == 0 and dz.data_ptr() % 16 == 0):
        M = x.shape[0]
        # the slab kernels are off the critical path (nothing downstream in backward reads them): launch them
        # on the sink's side stream so they overlap the latency-bound dgrad chain; joined in sink.flush()
        cur = torch.cuda.current_stream()
        ctxm = torch.cuda.stream(sink.side) if sink.side is not None else contextlib.nullcontext()
        if sink.side is not None:
            sink.side.wait_stream(cur)
            sink.keep.append((x, V, dz, U))
        with ctxm:
            if x.dtype == torch.bfloat16 and r in (16, 32, 48, 64):
                # dA and dB in one launch (matrix-core slab kernel), or postponed to the sink's batch launch
                defer = sink.will_defer(x, dz)
                rpa, nsa = LoraGradSink.plan_deferred(M) if defer else LoraGradSink.plan(M, x.shape[1])
                rpb_, nsb = LoraGradSink.plan_deferred(M) if defer else LoraGradSink.plan(M, dz.shape[1])
                wsA, wsB = sink.workspace(A, nsa), sink.workspace(B, nsb)
                sink.rank_pair(defer, M, r, x.shape[1], x, V, wsA, rpa, dz.shape[1], dz, U, wsB, rpb_)
                sink.add(wsA, gA, A.numel(), nsa)
                sink.add(wsB, gB, B.numel(), nsb)
            else:
                for Wd, Rk, P, g, tr in ((x, V, A, gA, False), (dz, U, B, gB, True)):
                    rpb, ns = LoraGradSink.plan(M, Wd.shape[1])
                    ws = sink.workspace(P, ns)
                    check(lib().cvft_lora_rank_partial(dt(Wd), M, Wd.shape[1], r, ptr(Wd), Wd.stride(0), ptr(Rk),
                                                       Rk.stride(0), ptr(ws), int(tr), rpb, stream()), "cvft_lora_rank_partial")
                    sink.add(ws, g, P.numel(), ns)
    else:
        rank_accum(x, V, gA, False)                               # dA[r,K] += V^T x
        rank_accum(dz, U, gB, True)                               # dB[N,r] += dz^T U
    if not direct:
        dA, dB = gA, gB
    return dA, dB


def _lin_bwd(x, U, ops, A_ref, B_ref, pack: LinearPack, scale: float, dz, need_dx: bool, need_dAB: bool,
             dx_residual=None, dact_src=None, dact: Optional[str] = None, drop=None, odrop=None, out_scale: float = 1.0):
    """Backward of one LoRA linear given dz = gradient at its pre-activation output.
    dx = (dz W + (scale * dz B) A) [* act'(dact_src)] [* keep(odrop) / (1 - p)] [+ dx_residual]  -- the bracketed links ride in
    the dgrad epilogue (fused producer-activation backward, the producer's dropout mask, fused gradient accumulation);
    dA / dB go to the parameters' flat .grad buffers (through the active LoraGradSink when there is one) or are returned."""
    dx = dA = dB = V = None
    has_lora = ops is not None
    # out_scale (a constant factor on dx, applied as the dgrad launch's alpha before act'): the keep scale of a producer whose
    # mask is already in dact_src (FeedForwardFn's ReLU form); not combined with dx_residual
    assert out_scale == 1.0 or dx_residual is None

    def producer_chain(dh):
        """act'(z) and the producer's dropout mask as passes of their own (launches whose epilogue could not take them)"""
        if out_scale != 1.0:
            dh = dh * out_scale
        if dact_src is None:
            return dh if odrop is None else dropout_raw(dh, odrop[0], odrop[1])
        if odrop is None:
            return act_bwd(dact_src, dh, dact)
        out = torch.empty_like(dact_src)
        check(lib().cvft_act_dropout(dt(dact_src), dact_src.numel(), ACT[dact], ptr(dact_src), ptr(_c(dh)), ptr(out), float(odrop[0]),
                                     ptr(_DROPOUT["seed"]), int(odrop[1]), stream()), "cvft_act_dropout")
        return out
    if has_lora and drop is not None:
        # lora_dropout: dx = dz W + mask/(1-p) * (V A); the masked term is added inside the GEMM launch (xdrop: it joins the
        # accumulators BEFORE the epilogue, so act'(z) / the producer's mask still ride there) or by the side-dgrad kernel
        # (then the producer chain runs as a pass of its own); the adapter gradients see drop(x), re-materialised from the site
        Ac, At, Bc, Bt = ops
        V = _side_v(dz, Bt, scale)
        if need_dx:
            if _can_xdrop(dz, pack.Wb, V, At, dx_residual):
                dx = gemm(dz, pack.Wb, U=V, Bl=At, residual=dx_residual, xdrop=(drop[0], [drop[1]] * (V.shape[1] // 16)),
                          dact_src=dact_src, dact=dact, odrop=odrop, alpha=out_scale)
            else:
                assert dx_residual is None or (dact_src is None and odrop is None)
                dx = gemm(dz, pack.Wb, residual=dx_residual)
                dx = producer_chain(side_dgrad(V, Ac, dx, drop[0], [drop[1]]))
        if need_dAB:
            dA, dB = _lora_param_grads(dropped_input(x, drop[0], drop[1]), U, V, dz, A_ref, B_ref, ops)
        return dx, dA, dB
    if has_lora:
        Ac, At, Bc, Bt = ops
        V = _take_side_v(dz, Bt, scale)
        if V is not None:
            pass
        elif need_dx and _can_fuse(dz, Bt, At, pack.K, pack.N):
            # dgrad with the side path fused: V = s * dz B is produced by the same launch
            V = torch.empty((dz.shape[0], Bt.shape[0]), dtype=dz.dtype, device=dz.device)
            dx = gemm(dz, pack.Wb, La=Bt, lora_scale=scale, Uout=V, Bl=At, dact_src=dact_src, dact=dact, residual=dx_residual,
                      odrop=odrop, alpha=out_scale)
        elif need_dx or need_dAB:
            V = gemm(dz, Bt, alpha=scale)                         # [M, r] = s * dz B
    if need_dx and dx is None:
        dx = _mm(dz, pack, 'b', pack.Wb, U=V, Bl=None if V is None else ops[1], dact_src=dact_src, dact=dact, residual=dx_residual,
                 odrop=odrop, alpha=out_scale)
    if has_lora and need_dAB:
        dA, dB = _lora_param_grads(x, U, V, dz, A_ref, B_ref, ops)
    return dx, dA, dB


class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b + scale * (x A^T) B^T) (+ residual): reference lora.py:64-76 as ONE
    GEMM launch plus a rank-r pre-GEMM; backward = dgrad (+ rank-r side path) and dA/dB, the latter
    accumulated straight into the parameters' (flat) .grad buffers when those exist."""

    @staticmethod
    def forward(ctx, x, A, B, residual, pack: LinearPack, scale: float, act: Optional[str], drop_p: float = 0.0,
                out_drop_p: float = 0.0):
        x = _c(x)
        need_grad = any(ctx.needs_input_grad[:3])
        ctx.padded = False
        ctx.drop = [float(drop_p), _next_drop_site()] if (drop_p > 0 and A is not None) else None
        # y = residual + dropout(linear(x)) (encoder_layer.py:95 / 104) with the mask applied in the GEMM epilogue; backward
        # re-derives it from the site (one cvft_dropout_add pass over dy)
        ctx.odrop = (float(out_drop_p), _next_drop_site()) if out_drop_p > 0 else None
        if A is None and act is None and residual is None and pack.Npad != pack.N and ctx.odrop is None:
            Wf, bias, _ = pack.padded
            y = gemm(x, Wf, bias=bias)                                  # [M][Npad], pad columns exactly zero
            _register_zero_padded(y)
            ctx.pack, ctx.padded = pack, True
            ctx.save_for_backward(x, None, None)
            return y[:, :pack.N]
        y, U, z, ops = _lin_fwd(x, A, B, pack, scale, act, residual, need_grad, ctx.drop, ctx.odrop)
        _note_out_drop(y, ctx.odrop, None if (ops is None or act) else (ops[3], scale))
        ctx.pack, ctx.scale, ctx.act = pack, scale, act
        ctx.ops, ctx.A_ref, ctx.B_ref = ops, A, B
        ctx.save_for_backward(x, U, z)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, U, z = ctx.saved_tensors
        if ctx.padded:
            dx = gemm(_widen_zero_padded(dy, ctx.pack.Npad), ctx.pack.padded[2]) if ctx.needs_input_grad[0] else None
            return dx, None, None, None, None, None, None, None, None
        dy = _c(dy) if (ctx.act or ctx.odrop) else _rowc(dy)
        dres = dy if ctx.needs_input_grad[3] else None
        if ctx.odrop is not None:
            dy = _masked_dy(dy, ctx.odrop)
        dz = act_bwd(z, dy, ctx.act) if ctx.act else dy
        dx, dA, dB = _lin_bwd(x, U, ctx.ops, ctx.A_ref, ctx.B_ref, ctx.pack, ctx.scale, dz, ctx.needs_input_grad[0],
                              ctx.needs_input_grad[1] or ctx.needs_input_grad[2], drop=ctx.drop)
        return dx, dA, dB, dres, None, None, None, None, None


class LinearQKVFn(torch.autograd.Function):
    """q, k, v = three LoRA linears of the SAME input (attention projections).  Forward is three launches; the
    point is backward: dx = dq Wq + dk Wk + dv Wv is accumulated through the dgrad epilogue (each launch adds the
    previous partial), so autograd never sees three separate dx tensors to sum."""

    @staticmethod
    def forward(ctx, x, Aq, Bq, Ak, Bk, Av, Bv, packs, scales):
        x = _c(x)
        outs, saved, ctx.ops = [], [x], []
        for (A, B), pack, s in zip(((Aq, Bq), (Ak, Bk), (Av, Bv)), packs, scales):
            y, U, _, ops = _lin_fwd(x, A, B, pack, s, None, None, False)
            outs.append(y)
            saved.append(U)
            ctx.ops.append(ops)
        ctx.packs, ctx.scales, ctx.refs = packs, scales, ((Aq, Bq), (Ak, Bk), (Av, Bv))
        ctx.save_for_backward(*saved)
        return tuple(outs)

    @staticmethod
    def backward(ctx, dq, dk, dv):
        x, Uq, Uk, Uv = ctx.saved_tensors
        dx = None
        grads = []
        for i, (d, U) in enumerate(((dq, Uq), (dk, Uk), (dv, Uv))):
            A, B = ctx.refs[i]
            if d is None:
                grads += [None, None]
                continue
            need_dAB = ctx.needs_input_grad[1 + 2 * i] or ctx.needs_input_grad[2 + 2 * i]
            dxi, dA, dB = _lin_bwd(x, U, ctx.ops[i], A, B, ctx.packs[i], ctx.scales[i], _rowc(d), ctx.needs_input_grad[0],
                                   need_dAB, dx_residual=dx)
            dx = dxi if dxi is not None else dx
            grads += [dA, dB]
        return (dx, *grads, None, None)


class QKVStack:
    """Frozen side of a stacked q|k|v projection: W [3N, K], its transpose [K, 3N], bias [3N] (compute dtype)."""

    def __init__(self, packs):
        self.N, self.K = packs[0].N, packs[0].K
        self.Wf = torch.cat([p.Wf for p in packs], 0).contiguous()
        self.Wb = self.Wf.t().contiguous()
        if all(p.bias is None for p in packs):
            self.bias = None
        else:
            self.bias = torch.cat([p.bias if p.bias is not None else torch.zeros(p.N, device=self.Wf.device) for p in packs]).contiguous()


class LinearQKVStackedFn(torch.autograd.Function):
    """q|k|v as ONE projection: Y [M, 3N] = x Wqkv^T + b + (s x A_stack^T) B_blk^T with the three adapters stacked
    (A_stack [3r, K]) / block-diagonal (B_blk [3N, 3r]).  Backward = one rank-3r GEMM, one dgrad over K = 3N, and two
    matrix-core slab launches (r = 3r) whose sub-blocks are routed to the six .grad buffers by the reduce tasks.
    bf16 training path only (needs the optimiser-maintained stacked shadows); anything else uses LinearQKVFn."""

    @staticmethod
    def forward(ctx, x, Aq, Bq, Ak, Bk, Av, Bv, wstack: QKVStack, ops, scale: float, drop_p: float = 0.0):
        x = _c(x)
        A, At, Bb, Bbt = ops
        ctx.drop = None
        if drop_p > 0:                    # lora_dropout: three mask sites (each LoRALinear owns its nn.Dropout)
            pre = take_pre_u(x, A, scale, drop_p, 3)          # made by the LayerNorm launch that produced x?
            if pre is not None:
                U, sites = pre
            else:
                sites = [_next_drop_site() for _ in range(3)]
                U = skinny_dropout(x, A, scale, float(drop_p), sites, keep_dropped=True)
            ctx.drop = (float(drop_p), sites)
            Y = gemm(x, wstack.Wf, bias=wstack.bias, U=U, Bl=Bb)
        elif QKV_FUSE_SIDE and x.shape[1] % 64 == 0 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0:
            U = torch.empty((x.shape[0], A.shape[0]), dtype=x.dtype, device=x.device)     # [M, 3r], written by the launch
            Y = gemm(x, wstack.Wf, bias=wstack.bias, La=A, lora_scale=scale, Uout=U, Bl=Bb)
        else:
            U = gemm(x, A, alpha=scale)
            Y = _mm(x, wstack, 'f', wstack.Wf, bias=wstack.bias, U=U, Bl=Bb)       # [M, 3N]
        ctx.w, ctx.ops, ctx.scale, ctx.refs = wstack, ops, scale, ((Aq, Bq), (Ak, Bk), (Av, Bv))
        ctx.save_for_backward(x, U)
        N = wstack.N
        return Y[:, :N], Y[:, N:2 * N], Y[:, 2 * N:]

    @staticmethod
    def backward(ctx, dq, dk, dv):
        x, U = ctx.saved_tensors
        w, (A, At, Bb, Bbt), scale = ctx.w, ctx.ops, ctx.scale
        N, M = w.N, x.shape[0]
        es = dq.element_size()
        if (dq.stride() == (3 * N, 1) and dk.stride() == (3 * N, 1) and dv.stride() == (3 * N, 1)
                and dk.data_ptr() == dq.data_ptr() + N * es and dv.data_ptr() == dq.data_ptr() + 2 * N * es):
            dY = torch.as_strided(dq, (M, 3 * N), (3 * N, 1), dq.storage_offset())     # the attention backward's fused buffer
        else:
            dY = torch.cat([dq, dk, dv], 1)
        if ctx.drop is not None:
            return LinearQKVStackedFn._backward_dropout(ctx, x, U, dY)
        if (QKV_FUSE_SIDE and ctx.needs_input_grad[0] and dY.shape[1] % 64 == 0 and dY.stride(0) % 8 == 0
                and dY.data_ptr() % 16 == 0):
            V = torch.empty((M, Bbt.shape[0]), dtype=dY.dtype, device=dY.device)
            dx = gemm(dY, w.Wb, La=Bbt, lora_scale=scale, Uout=V, Bl=At)   # V = s * dY B_blk produced by the same launch
        else:
            V = gemm(dY, Bbt, alpha=scale)                              # [M, 3r] = s * dY B_blk
            dx = _mm(dY, w, 'b', w.Wb, U=V, Bl=At) if ctx.needs_input_grad[0] else None
        r3 = V.shape[1]
        r = r3 // 3
        sink = LoraGradSink.active
        grads = [(a.grad, b.grad) for a, b in ctx.refs]
        direct = all(g is not None and g.dtype == torch.float32 and g.is_contiguous() for pair in grads for g in pair)
        if sink is not None and direct and sink.side is None:
            K = x.shape[1]
            defer = r3 in (16, 32, 48, 64) and sink.will_defer(x, dY)
            rpa, nsa = LoraGradSink.plan_deferred(M) if defer else LoraGradSink.plan(M, K)
            rpb_, nsb = LoraGradSink.plan_deferred(M) if defer else LoraGradSink.plan(M, 3 * N)
            wsA = sink.workspace(ctx.refs[0][0], nsa * 3)       # slab [3r, K] per row block
            wsB = sink.workspace(ctx.refs[0][1], nsb * 9)       # slab [3N, 3r] per row block
            sink.rank_pair(defer, M, r3, K, x, V, wsA, rpa, 3 * N, dY, U, wsB, rpb_)
            for i, (gA, _) in enumerate(grads):
                sink.add_block(wsA.data_ptr() + i * r * K * 4, gA, r, K, K, r3 * K, nsa)
            for i, (_, gB) in enumerate(grads):
                sink.add_block(wsB.data_ptr() + (i * N * r3 + i * r) * 4, gB, N, r, r3, 3 * N * r3, nsb)
            out = [None] * 6
        else:
            out = []
            for i, (gA, gB) in enumerate(grads):
                tA = gA if direct else torch.zeros((r, x.shape[1]), dtype=torch.float32, device=x.device)
                tB = gB if direct else torch.zeros((N, r), dtype=torch.float32, device=x.device)
                rank_accum(x, V[:, i * r:(i + 1) * r], tA, False)
                rank_accum(dY[:, i * N:(i + 1) * N], U[:, i * r:(i + 1) * r], tB, True)
                out += [None, None] if direct else [tA, tB]
        return (dx, *out, None, None, None, None)

    @staticmethod
    def _backward_dropout(ctx, x, U, dY):
        w, (A, At, Bb, Bbt), scale = ctx.w, ctx.ops, ctx.scale
        p, sites = ctx.drop
        N, M, K = w.N, x.shape[0], x.shape[1]
        V = gemm(dY, Bbt, alpha=scale)                                  # [M, 3r]
        dx = None
        if ctx.needs_input_grad[0]:
            if V.shape[1] == 16 * len(sites) and _can_xdrop(dY, w.Wb, V, At, None):
                dx = gemm(dY, w.Wb, U=V, Bl=At, xdrop=(p, list(sites)))  # + sum_t mask_t/(1-p) (V_t A_t), inside the launch
            else:
                dx = side_dgrad(V, A, gemm(dY, w.Wb), p, sites)
        r3 = V.shape[1]
        r = r3 // 3
        sink = LoraGradSink.active
        grads = [(a.grad, b.grad) for a, b in ctx.refs]
        direct = all(g is not None and g.dtype == torch.float32 and g.is_contiguous() for pair in grads for g in pair)
        xds = [dropped_input(x, p, st) for st in sites]                 # the three dropped inputs (written by the forward kernel)
        if sink is not None and direct and sink.side is None:
            defer = STACKED_DROP_DEFER and r == 16 and r3 == 48 and sink.will_defer(x, dY)
            rpa, nsa = LoraGradSink.plan_deferred(M) if defer else LoraGradSink.plan(M, K)
            rpb_, nsb = LoraGradSink.plan_deferred(M) if defer else LoraGradSink.plan(M, 3 * N)
            wsB = sink.workspace(ctx.refs[0][1], nsb * 9)
            keep = [sink.workspace(ctx.refs[i][0], nsa) for i in range(3)]
            if defer:                       # the four slab products join the sink's end-of-backward batch launches
                for i in range(3):
                    sink.defer_one(M, r, K, xds[i], V[:, i * r:(i + 1) * r], keep[i], 0, rpa, keep=(V,))
                sink.defer_one(M, r3, 3 * N, dY, U, wsB, 1, rpb_)
            else:
                probs = (cb.RankProb * 3)()
                for i in range(3):
                    Vi = V[:, i * r:(i + 1) * r]
                    probs[i].C, probs[i].Wd, probs[i].ldw = K, xds[i].data_ptr(), xds[i].stride(0)
                    probs[i].Rk, probs[i].ldr, probs[i].part = Vi.data_ptr(), V.stride(0), keep[i].data_ptr()
                    probs[i].transpose_out, probs[i].rows_per_block = 0, rpa
                check(lib().cvft_lora_rank_partial_multi(M, r, 3, probs, stream()), "cvft_lora_rank_partial_multi")
                check(lib().cvft_lora_rank_partial(dt(dY), M, 3 * N, r3, ptr(dY), dY.stride(0), ptr(U), U.stride(0), ptr(wsB), 1, rpb_,
                                                   stream()), "cvft_lora_rank_partial")
            for i, (gA, _) in enumerate(grads):
                sink.add(keep[i], gA, gA.numel(), nsa)
            for i, (_, gB) in enumerate(grads):
                sink.add_block(wsB.data_ptr() + (i * N * r3 + i * r) * 4, gB, N, r, r3, 3 * N * r3, nsb)
            sink.keep.append((xds, V, U, dY))
            out = [None] * 6
        else:
            out = []
            for i, (gA, gB) in enumerate(grads):
                tA = gA if direct else torch.zeros((r, K), dtype=torch.float32, device=x.device)
                tB = gB if direct else torch.zeros((N, r), dtype=torch.float32, device=x.device)
                rank_accum(xds[i], V[:, i * r:(i + 1) * r], tA, False)
                rank_accum(dY[:, i * N:(i + 1) * N], U[:, i * r:(i + 1) * r], tB, True)
                out += [None, None] if direct else [tA, tB]
        return (dx, *out, None, None, None, None)


class FeedForwardFn(torch.autograd.Function):
    """y = W2 drop_in(act(W1 x + b1)) + b2, then residual + drop_out(y): both linears with optional LoRA (and lora_dropout), the
    encoder's inner and residual-branch dropouts (positionwise_feed_forward.py:54, encoder_layer.py:104 / 234) optional.
    Forward: the activation, the inner mask and the pre-activation save ride in W1's epilogue, the outer mask and the residual
    in W2's.  Backward applies act'(z) and the inner mask in the epilogue of W2's dgrad launch -- no separate pass over the
    [M, hidden] tensor either way (at the LLM's 5 328 x 4 096 that was 130 MB of traffic per layer and step)."""

    @staticmethod
    def forward(ctx, x, A1, B1, A2, B2, residual, pack1, pack2, s1: float, s2: float, act: str, p1: float = 0.0, p2: float = 0.0,
                p_in: float = 0.0, p_out: float = 0.0):
        x = _c(x)
        need_grad = any(ctx.needs_input_grad[:5])
        # mask sites in the order the unfused path draws them: W1's adapter, inner, W2's adapter, outer
        d1 = [float(p1), _next_drop_site()] if (p1 > 0 and A1 is not None) else None
        od_in = (float(p_in), _next_drop_site()) if p_in > 0 else None
        d2 = [float(p2), _next_drop_site()] if (p2 > 0 and A2 is not None) else None
        od_out = (float(p_out), _next_drop_site()) if p_out > 0 else None
        ctx.drops = (d1, d2, od_in, od_out)
        # ReLU: relu'(z) * keep_in / (1 - p_in) == (h > 0) / (1 - p_in) with h = drop_in(relu(z)), which is saved for dA2 anyway --
        # no pre-activation copy in forward (43 MB per LLM layer), no inner-mask draws in backward
        ctx.relu_h = RELU_FROM_H and act == "relu" and A2 is not None
        h, U1, z, ops1 = _lin_fwd(x, A1, B1, pack1, s1, act, None, need_grad and not ctx.relu_h, d1, od_in)
        y, U2, _, ops2 = _lin_fwd(h, A2, B2, pack2, s2, None, residual, False, d2, od_out)
        _note_out_drop(y, od_out, None if ops2 is None else (ops2[3], s2))
        ctx.cfg = (pack1, pack2, s1, s2, act, ops1, ops2, (A1, B1), (A2, B2))
        ctx.save_for_backward(x, U1, z, h if A2 is not None else None, U2)      # (h: dA2 = V2^T h, or drop(h) re-derived)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, U1, z, h, U2 = ctx.saved_tensors
        pack1, pack2, s1, s2, act, ops1, ops2, (A1, B1), (A2, B2) = ctx.cfg
        d1, d2, od_in, od_out = ctx.drops
        dy = _c(dy) if od_out is not None else _rowc(dy)
        dres = dy if ctx.needs_input_grad[5] else None
        if od_out is not None:
            dy = _masked_dy(dy, od_out)
        need1 = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        need2 = ctx.needs_input_grad[3] or ctx.needs_input_grad[4]
        need_dx = ctx.needs_input_grad[0]
        if ctx.relu_h:
            dz, dA2, dB2 = _lin_bwd(h, U2, ops2, A2, B2, pack2, s2, dy, need_dx or need1, need2, dact_src=h, dact=act, drop=d2,
                                    out_scale=1.0 / (1.0 - od_in[0]) if od_in is not None else 1.0)
        else:
            dz, dA2, dB2 = _lin_bwd(h, U2, ops2, A2, B2, pack2, s2, dy, need_dx or need1, need2, dact_src=z, dact=act, drop=d2,
                                    odrop=od_in)
        dx = dA1 = dB1 = None
        if dz is not None:
            dx, dA1, dB1 = _lin_bwd(x, U1, ops1, A1, B1, pack1, s1, dz, need_dx, need1, drop=d1)
        return dx, dA1, dB1, dA2, dB2, dres, None, None, None, None, None, None, None, None, None


class LoraSideFn(torch.autograd.Function):
    """y = base + scale * (xd A^T) B^T -- the LoRA side path on its own input (training with lora_dropout > 0, where
    xd = dropout(x) differs from the main path's x; lora.py:70-73).  Two rank-r products, no zero-weight main GEMM."""

    @staticmethod
    def forward(ctx, xd, A, B, base, scale: float):
        xd = _c(xd)
        ops = (*_lora_operands(A, xd.dtype), *_lora_operands(B, xd.dtype))
        U = gemm(xd, ops[0], alpha=scale)                       # [M, r]
        y = gemm(U, ops[2], residual=_c(base))                  # [M, N], K = r
        ctx.ops, ctx.scale, ctx.refs = ops, scale, (A, B)
        ctx.save_for_backward(xd, U)
        return y

    @staticmethod
    def backward(ctx, dy):
        xd, U = ctx.saved_tensors
        Ac, At, Bc, Bt = ctx.ops
        dy = _rowc(dy)
        V = gemm(dy, Bt, alpha=ctx.scale)                       # [M, r] = s dy B
        dxd = gemm(V, At) if ctx.needs_input_grad[0] else None  # [M, K]
        dA = dB = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dA, dB = _lora_param_grads(xd, U, V, dy, ctx.refs[0], ctx.refs[1], ctx.ops)
        return dxd, dA, dB, (dy if ctx.needs_input_grad[3] else None), None


def lora_side(xd, A, B, base, scale: float):
    return LoraSideFn.apply(xd, A, B, base, scale)


def lora_linear(x, pack: LinearPack, A=None, B=None, scale: float = 1.0, act: Optional[str] = None, residual=None,
                drop_p: float = 0.0, out_drop_p: float = 0.0):
    """drop_p > 0 (train mode, lora.py:70): the side path sees dropout(x); needs _can_drop_fuse(x, r).
    out_drop_p > 0: y = residual + dropout(act(linear(x))), mask in the GEMM epilogue (N % 4 == 0)."""
    return LinearFn.apply(x, A, B, residual, pack, scale, act, drop_p, out_drop_p)


# residual + dropout(linear(.)) with the mask in the GEMM epilogue instead of a cvft_dropout_add pass (CVFT_OUT_DROP_FUSE=0: A/B)
OUT_DROP_FUSE = _os.environ.get("CVFT_OUT_DROP_FUSE", "1") != "0"
# train-mode feed-forward as ONE Function (activation + inner mask in W1's epilogue, act' + inner mask in W2's dgrad epilogue)
FFN_TRAIN_FUSE = _os.environ.get("CVFT_FFN_TRAIN_FUSE", "1") != "0"
# ReLU feed-forward backward from the saved (dropped) hidden activations instead of a pre-activation copy (FeedForwardFn)
RELU_FROM_H = _os.environ.get("CVFT_RELU_FROM_H", "1") != "0"

QKV_STACKING = _os.environ.get("CVFT_QKV_STACK", "1") != "0"
# rank-48 side products inside the main launches: measured slower than the dedicated skinny kernel (39.7 vs 39.0 ms/step:
# the extra MFMAs land on half of the block's waves), kept selectable
QKV_FUSE_SIDE = _os.environ.get("CVFT_QKV_FUSE", "0") != "0"


def _qkv_stacked_operands(x, packs, loras, scales):
    """(QKVStack, stacked LoRA operands) when the stacked bf16 path applies, else None."""
    if not QKV_STACKING or x.dtype != torch.bfloat16 or any(a is None or b is None for a, b in loras):
        return None
    (Aq, Bq), (Ak, Bk), (Av, Bv) = loras
    if not (scales[0] == scales[1] == scales[2] and Aq.shape == Ak.shape == Av.shape and Bq.shape == Bk.shape == Bv.shape
            and Aq.shape[0] == 16 and packs[0].N == packs[1].N == packs[2].N and packs[0].N % 8 == 0 and packs[0].K % 8 == 0):
        return None
    ref = getattr(Aq, "_cvft_opt", None)
    opt = ref() if ref is not None else None
    if opt is None:
        return None
    for P in (Aq, Bq, Ak, Bk, Av, Bv):                    # masters edited in place since the last shadow refresh
        if getattr(P, "_cvft_shadow_ver", -1) != P._version:
            return None
    ops = opt.stack_for((Aq, Ak, Av), (Bq, Bk, Bv))
    if ops is None:
        return None
    # the stacked weight lives ON the first pack (with the two partners it was built from, compared by identity): a module-level
    # table keyed by id(pack) handed a later model the stack of a freed one whose ids had been recycled (wrong-shape assertion in
    # tests/test_model_gpu.py when it ran behind tests that build and drop other models)
    ent = packs[0].__dict__.get("_cvft_qkv_stack")
    if ent is None or ent[0] is not packs[1] or ent[1] is not packs[2]:
        ent = (packs[1], packs[2], QKVStack(packs))
        packs[0].__dict__["_cvft_qkv_stack"] = ent
    return ent[2], ops


def lora_linear_qkv(x, packs, loras, scales, drop_p: float = 0.0):
    """(q, k, v) = three LoRA linears of x; loras = ((Aq, Bq), (Ak, Bk), (Av, Bv)) with None entries for plain layers.
    drop_p > 0: lora_dropout (train mode) -- only on the stacked bf16 path; returns None when that path does not apply
    (the caller then falls back to three separate dropout-path linears)."""
    (Aq, Bq), (Ak, Bk), (Av, Bv) = loras
    st = _qkv_stacked_operands(x, packs, loras, scales)
    if drop_p > 0:
        if st is None or not _can_drop_fuse(x, 48):
            return None
        return LinearQKVStackedFn.apply(x, Aq, Bq, Ak, Bk, Av, Bv, st[0], st[1], scales[0], drop_p)
    if st is not None:
        return LinearQKVStackedFn.apply(x, Aq, Bq, Ak, Bk, Av, Bv, st[0], st[1], scales[0], 0.0)
    return LinearQKVFn.apply(x, Aq, Bq, Ak, Bk, Av, Bv, tuple(packs), tuple(scales))


def lora_feed_forward(x, pack1, pack2, lora1=(None, None), lora2=(None, None), s1: float = 1.0, s2: float = 1.0,
                      act: str = "relu", residual=None, p1: float = 0.0, p2: float = 0.0, p_in: float = 0.0, p_out: float = 0.0):
    """p1 / p2: lora_dropout of the two adapters; p_in / p_out: the encoder's inner and residual-branch dropouts."""
    return FeedForwardFn.apply(x, lora1[0], lora1[1], lora2[0], lora2[1], residual, pack1, pack2, s1, s2, act, p1, p2, p_in, p_out)


# ---------------------------------------------------------------------------------
# estimator transformer block as row-tile chain kernels (csrc/block_fused.hip)
# ---------------------------------------------------------------------------------
class BlockTailFn(torch.autograd.Function):
    """out = x1 + W2 gelu(W1 LN(x1) + b1) + b2 with x1 = x0 + o Wo^T + bo (o given) or x1 = x0 (o None): the second half of a
    BasicTransformerBlock (matcha transformer.py:290-316 == modules.py:362-375) as ONE launch each way.  No LoRA on to_out /
    ff.net.* in the flow target list (config.py), so backward is input gradients only."""

    @staticmethod
    def forward(ctx, o, x0, pack, act: str, link=None):
        x0 = _c(x0)
        M = x0.shape[0]
        a = cb.BlockTailArgs()
        a.M = M
        x1 = x0
        assert (o is not None) == (pack.DI > 0), "BlockTailPack built with / without to_out must match the call"
        mode = block_tail_lean()
        if mode >= 2 and pack.F < 256:                 # (the 64-row forms' rounds lag by one: they need two of them)
            mode = 1
        # every form saves gelu'(z) in the same [row tile][hidden tile][64 lanes][16] order, so the two directions may use different
        # forms: 3 = 32-row forward (the faster forward) + 64-row backward (same latency as the 32-row one on half the CUs)
        if mode == 4 and pack.W_fwd_wide8 is None:     # (the eight-wave forward: F % 256 == 0, 512 <= F <= 1024)
            mode = 3
        ctx.lean = 2 if mode >= 3 else mode            # (backward form)
        fwd_form = 0 if mode == 3 else mode
        a.DI, a.W_fwd, a.lean = pack.DI, ptr({0: pack.W_fwd, 1: pack.W_fwd_lean, 2: pack.W_fwd_wide, 4: pack.W_fwd_wide8}[fwd_form]), int(fwd_form)
        if o is not None:
            assert o.shape == (M, pack.DI) and o.stride(1) == 1
            x1 = torch.empty_like(x0)
            a.o, a.ldo, a.x0, a.bo = ptr(o), o.stride(0), ptr(x0), ptr(pack.bo)
        need = (o is not None and ctx.needs_input_grad[0]) or ctx.needs_input_grad[1]
        out = torch.empty_like(x0)
        mean = torch.empty(M, dtype=torch.float32, device=x0.device)
        rstd = torch.empty(M, dtype=torch.float32, device=x0.device)
        # (the wide form always stores z and works on whole 64-row groups)
        z = torch.empty(-(-M // 64) * 64 * pack.F, dtype=x0.dtype, device=x0.device) if (need or fwd_form in (2, 4)) else None
        a.x1, a.gamma, a.beta, a.eps = ptr(x1), ptr(pack.gamma), ptr(pack.beta), pack.eps
        a.b1, a.F, a.b2, a.act = ptr(pack.b1), pack.F, ptr(pack.b2), ACT[act]
        a.z, a.mean, a.rstd, a.out = ptr(z), ptr(mean), ptr(rstd), ptr(out)
        att = _take_hand(o, _H_ATTN_O) if o is not None else None
        head = None
        if link is not None and fwd_form == 0 and o is not None and pack.DI == 512 and 256 <= pack.F <= 1024:
            # the NEXT block's norm1 + q|k|v head rides in the same launch (csrc/block_fused.hip, block_link_fwd_kernel); its products
            # wait on `out` for that block's BlockQkvFn
            hpack, hops, hscale, hp, lpack, hneed = link
            head = _QkvHead(out, hpack, hops, hscale, hp, hneed, wide=False)
            with _Bracket("block_link_fwd", 2.0 * M * 256 * (2 * pack.F + pack.DI + hpack.N3 + 48) + 2.0 * M * 48 * 512,
                          2.0 * (M * (3 * 256 + pack.DI + hpack.N3) + 256 * (2 * pack.F + pack.DI + hpack.N3))):
                check(lib().cvft_block_link_fwd(C.byref(a), C.byref(head.a), ptr(lpack.W_fwd), stream()), "cvft_block_link_fwd")
            head.a = None
            head.tail_rec = _LinkRec(lpack)
            head.tail_rec.tail_ctx = ctx                # (one-way: head -> record -> this context, the direction of the autograd edges)
            _hand(out, _H_LINK, head)
        else:
            with _Bracket("block_tail_fwd", 2.0 * M * 256 * (2 * pack.F + (pack.DI if o is not None else 0)),
                          2.0 * (M * (3 * 256 + pack.DI) + 256 * (2 * pack.F + pack.DI))):
                check(lib().cvft_block_tail_fwd(C.byref(a), stream()), "cvft_block_tail_fwd")
        ctx.attn_dims = None
        if att is not None and att[1] * att[3] == M and att[2] * 64 == pack.DI and o.stride(0) % 8 == 0:
            ctx.attn_dims = att[1:]                     # (B, H, T): backward forms the attention backward's delta next to do
            ctx.save_for_backward(x1, z, mean, rstd, o, att[0])
        else:
            ctx.save_for_backward(x1, z, mean, rstd)
        ctx.pack, ctx.act, ctx.has_o = pack, act, o is not None
        return out

    @staticmethod
    def tail_bwd_args(ctx, dy, lean: int, want_do: bool):
        """-> (argument block, dx1, do, delta): delta [B, H, T] (or None) is written next to do when the forward saw the estimator
        attention's hand-over (`_H_ATTN_O`) -- forms 0 and 2 of the kernels"""
        x1, z, mean, rstd, *att = ctx.saved_tensors
        pack = ctx.pack
        M = x1.shape[0]
        a = cb.BlockTailBwdArgs()
        dx1 = torch.empty_like(x1)
        do = None
        a.M, a.x1, a.dy, a.gamma, a.mean, a.rstd, a.z = M, ptr(x1), ptr(dy), ptr(pack.gamma), ptr(mean), ptr(rstd), ptr(z)
        a.W_bwd, a.F, a.DI, a.act, a.dx1 = ptr((pack.W_bwd, pack.W_bwd_lean, pack.W_bwd_wide)[lean]), pack.F, pack.DI, ACT[ctx.act], ptr(dx1)
        a.lean = int(lean)
        delta = None
        if want_do:
            do = torch.empty((M, pack.DI), dtype=x1.dtype, device=x1.device)
            a.dout, a.lddo = ptr(do), do.stride(0)
            if att and ctx.attn_dims is not None and lean in (0, 2) and pack.DI == 512:
                B, H, T = ctx.attn_dims
                delta = torch.empty((B, H, T), dtype=torch.float32, device=x1.device)
                a.attn_o, a.attn_o_lo, a.ldao, a.delta, a.T = ptr(att[0]), ptr(att[1]), att[0].stride(0), ptr(delta), T
        return a, dx1, do, delta

    @staticmethod
    def backward(ctx, dy):
        rec = _take_hand(dy, _H_LINK_BWD)
        if rec is not None and rec.tail_ctx is ctx and rec.results is not None:
            dx1, do = rec.results                       # the next block's head backward ran this tail backward in its launch
            rec.results = rec.tail_ctx = None
            return do, dx1, None, None, None
        pack = ctx.pack
        dy = _c(dy)
        M = dy.shape[0]
        a, dx1, do, delta = BlockTailFn.tail_bwd_args(ctx, dy, ctx.lean, ctx.has_o and ctx.needs_input_grad[0])
        with _Bracket("block_tail_bwd", 2.0 * M * 256 * (2 * pack.F + (pack.DI if do is not None else 0)),
                      2.0 * (M * (3 * 256 + pack.F + pack.DI) + 256 * (2 * pack.F + pack.DI))):
            check(lib().cvft_block_tail_bwd(C.byref(a), stream()), "cvft_block_tail_bwd")
        if delta is not None:
            _hand(do, _H_DELTA, delta)
        return do, dx1, None, None, None


class _LinkRec:
    """One block boundary that ran as a linked launch in forward (BlockTailFn.forward -> _QkvHead.tail_rec -> BlockQkvFn's ctx):
    lets the head's backward find the tail's autograd context, and the tail's backward recognise its results."""

    def __init__(self, lpack):
        self.lpack, self.tail_ctx, self.results = lpack, None, None


class _QkvHead:
    """Outputs and argument block of one cvft_block_qkv_fwd-shaped launch on x [M, 256]: built by BlockQkvFn.forward for its own
    launch, or by the PREVIOUS block's linked tail launch (BlockTailFn.forward), which leaves it on the block output (_H_LINK)."""

    def __init__(self, x, pack, ops, scale: float, drop_p: float, need: bool, wide: bool):
        M = x.shape[0]
        A, At, Bb, Bbt = ops
        N3 = pack.N3
        self.pack, self.ops, self.scale, self.p, self.need = pack, ops, float(scale), float(drop_p), bool(need)
        self.Y = torch.empty((M, N3), dtype=x.dtype, device=x.device)
        self.U = torch.empty((M, 48), dtype=x.dtype, device=x.device)
        self.mean = torch.empty(M, dtype=torch.float32, device=x.device)
        self.rstd = torch.empty(M, dtype=torch.float32, device=x.device)
        a = cb.BlockQkvArgs()
        a.M, a.x, a.gamma, a.beta, a.eps, a.mean, a.rstd = M, ptr(x), ptr(pack.gamma), ptr(pack.beta), pack.eps, ptr(self.mean), ptr(self.rstd)
        a.W_fwd, a.bias, a.N3 = ptr(pack.W_fwd), ptr(pack.bias), N3
        a.wide = int(wide)
        a.A, a.lda, a.Bb, a.ldb = ptr(A), A.stride(0), ptr(Bb), Bb.stride(0)
        a.alpha, a.p = float(scale), float(drop_p)
        self.xds, self.y, self.sites, self.tail_rec = [], None, None, None
        if drop_p > 0:
            self.sites = [_next_drop_site() for _ in range(3)]
            a.seed = ptr(_DROPOUT["seed"])
            for i, st in enumerate(self.sites):
                a.sites[i] = st
            if need:
                self.xds = [torch.empty_like(x) for _ in range(3)]
                for i, t in enumerate(self.xds):
                    a.xd[i] = t.data_ptr()
        elif need:
            self.y = torch.empty_like(x)
            a.y_out = ptr(self.y)
        a.U, a.ldu, a.Y, a.ldy = ptr(self.U), self.U.stride(0), ptr(self.Y), self.Y.stride(0)
        self.a = a

    def serves(self, x, pack, ops, scale: float, drop_p: float, need: bool) -> bool:
        return (self.pack is pack and len(self.ops) == len(ops) and all(a is b for a, b in zip(self.ops, ops)) and self.scale == float(scale)
                and self.p == float(drop_p) and (self.need or not need) and self.Y.shape[0] == x.shape[0] and self.Y.dtype == x.dtype)


class BlockQkvFn(torch.autograd.Function):
    """(x, q, k, v) with q|k|v = stacked LoRA projections of norm1(x): LayerNorm, the adapters' rank-side products under
    lora_dropout and the projection in ONE launch (cvft_block_qkv_fwd); backward = V, the dgrad with the masked side term and the
    LayerNorm backward joined with the residual branch's gradient in ONE launch (cvft_block_qkv_bwd).  The adapter gradients go
    through the LoraGradSink exactly as LinearQKVStackedFn's (dA_t = V_t^T drop_t(y), dB = dY^T U)."""

    @staticmethod
    def forward(ctx, x, Aq, Bq, Ak, Bk, Av, Bv, pack, ops, scale: float, drop_p: float):
        linked = _take_hand(x, _H_LINK)
        x = _c(x)
        M = x.shape[0]
        N3 = pack.N3
        need = any(ctx.needs_input_grad[:7])
        mode = block_qkv_wide()
        ctx.wide = mode in ("1", "both")                    # (backward form)
        ctx.link_rec = None
        if linked is not None and linked.serves(x, pack, ops, scale, drop_p, need):
            hd = linked                                     # the previous block's tail launch already ran this head on these rows
            ctx.link_rec = hd.tail_rec
        else:
            hd = _QkvHead(x, pack, ops, scale, drop_p, need, wide=mode in ("1", "both", "fwd"))
            with _Bracket("block_qkv_fwd", 2.0 * M * 256 * (N3 + 48) + 2.0 * M * 48 * 512, 2.0 * (M * (256 + N3) + 256 * N3)):
                check(lib().cvft_block_qkv_fwd(C.byref(hd.a), stream()), "cvft_block_qkv_fwd")
            hd.a = None
        Y, U, mean, rstd, xds, y = hd.Y, hd.U, hd.mean, hd.rstd, hd.xds, hd.y
        ctx.sites = hd.sites
        ctx.pack, ctx.ops, ctx.scale, ctx.p = pack, ops, float(scale), float(drop_p)
        ctx.refs = ((Aq, Bq), (Ak, Bk), (Av, Bv))
        ctx.nx = len(xds)
        ctx.save_for_backward(x, mean, rstd, U, y, *xds)
        N = N3 // 3
        return x.view_as(x), Y[:, :N], Y[:, N:2 * N], Y[:, 2 * N:]

    @staticmethod
    def backward(ctx, dres, dq, dk, dv):
        x, mean, rstd, U, y, *xds = ctx.saved_tensors
        pack, (A, At, Bb, Bbt), scale, p = ctx.pack, ctx.ops, ctx.scale, ctx.p
        M, N3 = x.shape[0], pack.N3
        N = N3 // 3
        es = dq.element_size()
        if (dq.stride() == (N3, 1) and dk.stride() == (N3, 1) and dv.stride() == (N3, 1)
                and dk.data_ptr() == dq.data_ptr() + N * es and dv.data_ptr() == dq.data_ptr() + 2 * N * es):
            dY = torch.as_strided(dq, (M, N3), (N3, 1), dq.storage_offset())      # the attention backward's fused buffer
        else:
            dY = torch.cat([dq, dk, dv], 1)
        V = torch.empty((M, 48), dtype=x.dtype, device=x.device)
        dx = torch.empty_like(x)
        a = cb.BlockQkvBwdArgs()
        a.M, a.dY, a.lddy, a.dres, a.x = M, ptr(dY), dY.stride(0), ptr(None if dres is None else _c(dres)), ptr(x)
        a.gamma, a.mean, a.rstd, a.W_bwd, a.N3 = ptr(pack.gamma), ptr(mean), ptr(rstd), ptr(pack.W_bwd_wide if ctx.wide else pack.W_bwd), N3
        a.wide = int(ctx.wide)
        a.At, a.ldat, a.Bbt, a.ldbt = ptr(At), At.stride(0), ptr(Bbt), Bbt.stride(0)
        a.alpha, a.p = scale, p
        if p > 0:
            a.seed = ptr(_DROPOUT["seed"])
            for i, st in enumerate(ctx.sites):
                a.sites[i] = st
        a.V, a.ldv, a.dx = ptr(V), V.stride(0), ptr(dx)
        rec = ctx.link_rec
        tctx = None if rec is None else rec.tail_ctx
        if (tctx is not None and block_link_bwd_on() and tctx.has_o and tctx.needs_input_grad[0] and tctx.pack.DI == 512
                and 256 <= tctx.pack.F <= 1024):
            # the PREVIOUS block's tail backward rides in this launch (csrc/block_fused.hip, block_link_bwd_kernel): dx is its dy
            a.wide, a.W_bwd = 0, ptr(pack.W_bwd)
            ta, dx1, do, tdelta = BlockTailFn.tail_bwd_args(tctx, dx, 0, True)
            tp = tctx.pack
            with _Bracket("block_link_bwd", 2.0 * M * N3 * (256 + 48) + 2.0 * M * 48 * 256 + 2.0 * M * 256 * (2 * tp.F + tp.DI),
                          2.0 * (M * (N3 + 6 * 256 + tp.F + tp.DI) + 256 * (N3 + 2 * tp.F + tp.DI))):
                check(lib().cvft_block_link_bwd(C.byref(a), C.byref(ta), ptr(rec.lpack.W_bwd), stream()), "cvft_block_link_bwd")
            if tdelta is not None:
                _hand(do, _H_DELTA, tdelta)
            rec.results = (dx1, do)
            _hand(dx, _H_LINK_BWD, rec)
        else:
            with _Bracket("block_qkv_bwd", 2.0 * M * N3 * (256 + 48) + 2.0 * M * 48 * 256, 2.0 * (M * (N3 + 3 * 256) + 256 * N3)):
                check(lib().cvft_block_qkv_bwd(C.byref(a), stream()), "cvft_block_qkv_bwd")
        if not any(ctx.needs_input_grad[1:7]):
            return (dx, None, None, None, None, None, None, None, None, None, None)
        # adapter gradients: dA_t = V_t^T drop_t(y) (three rank-16 products, or one rank-48 product on y when p == 0), dB = dY^T U
        r3, K = 48, x.shape[1]
        r = r3 // 3
        sink = LoraGradSink.active
        grads = [(a_.grad, b_.grad) for a_, b_ in ctx.refs]
        direct = all(g is not None and g.dtype == torch.float32 and g.is_contiguous() for pair in grads for g in pair)
        srcs = xds if ctx.nx == 3 else [y, y, y]
        if sink is not None and direct and sink.side is None:
            defer = sink.will_defer(x, dY)
            rpa, nsa = LoraGradSink.plan_deferred(M) if defer else LoraGradSink.plan(M, K)
            rpb_, nsb = LoraGradSink.plan_deferred(M) if defer else LoraGradSink.plan(M, N3)
            wsB = sink.workspace(ctx.refs[0][1], nsb * 9)
            keep = [sink.workspace(ctx.refs[i][0], nsa) for i in range(3)]
            if defer:
                for i in range(3):
                    sink.defer_one(M, r, K, srcs[i], V[:, i * r:(i + 1) * r], keep[i], 0, rpa, keep=(V,))
                sink.defer_one(M, r3, N3, dY, U, wsB, 1, rpb_)
            else:
                probs = (cb.RankProb * 3)()
                for i in range(3):
                    Vi = V[:, i * r:(i + 1) * r]
                    probs[i].C, probs[i].Wd, probs[i].ldw = K, srcs[i].data_ptr(), srcs[i].stride(0)
                    probs[i].Rk, probs[i].ldr, probs[i].part = Vi.data_ptr(), V.stride(0), keep[i].data_ptr()
                    probs[i].transpose_out, probs[i].rows_per_block = 0, rpa
                check(lib().cvft_lora_rank_partial_multi(M, r, 3, probs, stream()), "cvft_lora_rank_partial_multi")
                check(lib().cvft_lora_rank_partial(dt(dY), M, N3, r3, ptr(dY), dY.stride(0), ptr(U), U.stride(0), ptr(wsB), 1, rpb_,
                                                   stream()), "cvft_lora_rank_partial")
            for i, (gA, _) in enumerate(grads):
                sink.add(keep[i], gA, gA.numel(), nsa)
            for i, (_, gB) in enumerate(grads):
                sink.add_block(wsB.data_ptr() + (i * N * r3 + i * r) * 4, gB, N, r, r3, N3 * r3, nsb)
            sink.keep.append((srcs, V, U, dY))
            out = [None] * 6
        else:
            out = []
            for i, (gA, gB) in enumerate(grads):
                tA = gA if direct else torch.zeros((r, K), dtype=torch.float32, device=x.device)
                tB = gB if direct else torch.zeros((N, r), dtype=torch.float32, device=x.device)
                rank_accum(srcs[i], V[:, i * r:(i + 1) * r], tA, False)
                rank_accum(dY[:, i * N:(i + 1) * N], U[:, i * r:(i + 1) * r], tB, True)
                out += [None, None] if direct else [tA, tB]
        return (dx, *out, None, None, None, None)


def block_qkv(x, loras, pack, ops, scale: float, drop_p: float = 0.0):
    """-> (x_residual, q, k, v); loras = ((Aq, Bq), (Ak, Bk), (Av, Bv)) masters, ops = their stacked bf16 shadows (A, At, Bb, Bbt)."""
    (Aq, Bq), (Ak, Bk), (Av, Bv) = loras
    return BlockQkvFn.apply(x, Aq, Bq, Ak, Bk, Av, Bv, pack, ops, scale, drop_p)


def block_tail(o, x0, pack, act: str = "gelu_erf", link=None):
    """x0 [M, 256] bf16 residual stream, o [M, DI] attention output (or None: feed-forward half only).
    link = (head pack, stacked operands, scale, p, BlockLinkPack, need) of the NEXT block's q|k|v head: run it in the same launch."""
    return BlockTailFn.apply(o, x0, pack, act, link)


def can_block_tail(x: torch.Tensor, d_ff: int, d_inner: int) -> bool:
    return (BLOCK_FUSE and x.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[1] == 256 and d_ff % 128 == 0
            and d_ff <= 2048 and d_inner in (256, 512))


BLOCK_FUSE = _os.environ.get("CVFT_BLOCK_FUSE", "1") != "0"
# Form of the block-tail kernels: 0 = CU-owning 32-row form (block_fused.hip), 1 = CU-sharing 32-row form (block_lean.hip), 2 = 64-row form
# (block_wide.hip), 3 = 32-row forward + 64-row backward, 4 = eight-wave 64-row forward (block_wide8.hip) + 64-row backward, "auto" (default) = 3
# while >= 3 chains share the chip, else 0.  Measured, same-box
# A/Bs (DESIGN sections 11 / 12): joint 22.85 (0) / 22.99 (1) / 22.71 (2) / 23.2 -> 22.8 (3); flow_only 14.66 (0) / 15.0 (3) / 15.7 (2): the 64-row
# backward has the 32-row one's latency on half the CUs, the 64-row forward is 9 us longer -- fewer CUs only pay when the chip is shared
BLOCK_LEAN = _os.environ.get("CVFT_BLOCK_LEAN", "auto")


def block_tail_lean() -> int:
    """0: CU-owning 32-row form (block_fused.hip); 1: CU-sharing 32-row form (block_lean.hip); 2: 64-row form (block_wide.hip);
    3: 32-row forward + 64-row backward; 4: eight-wave 64-row forward (block_wide8.hip) + 64-row backward"""
    if BLOCK_LEAN in ("0", "1", "2", "3", "4"):
        return int(BLOCK_LEAN)
    return 3 if lib().cvft_concurrent_chains() >= 3 else 0
# The attention backward's delta = rowsum(dO . (O + O_lo)) formed by the launch that PRODUCES dO, the block tail's backward (it holds the
# dO tile it is about to store; cvft_block_tail_bwd_args.delta), so that neither attention backward role loads O / O_lo: per launch
# 34.8 -> 33.1 us at T = 250 and 98.3 -> 64.8 us at T = 500 (B = 16, tools/bench_attn.py).  "0" off.
TAIL_DELTA = _os.environ.get("CVFT_TAIL_DELTA", "1") != "0"
# The tail of block i and the head of block i + 1 in one launch (cvft_block_link_fwd; 32-row forms): "0" off, "1" / "auto" (default) on.
# Graph-timed per boundary at M = 2000 (tools/bench_link.py): tail + 64-row head 47.4 us, tail + 32-row head on two workgroups per row
# tile 40.4 us, linked 43.1 us -- a workgroup's time is its weight stream (2 MB at ~50 GB/s per CU), which the link does not shorten:
# it saves the launch boundary only.  Same-box A/B, two pairs of 40 steps: joint 21.02 / 21.02 -> 20.79 / 20.79 ms (the pair there is
# tail + 64-row head); flow_only 14.51 / 14.51 -> 14.60 / 14.60 forward alone (the pair there is the two-workgroup head), 14.57 / 14.54
# -> 14.46 / 14.49 with the backward linked too
BLOCK_LINK = _os.environ.get("CVFT_BLOCK_LINK", "auto")


def block_link_on() -> bool:
    return BLOCK_LINK != "0"


# ... and the same boundary backwards (cvft_block_link_bwd: block i + 1's head backward + block i's tail backward), wherever the
# forward ran linked: "0" off, "1" / "auto" (default) on.  Per boundary (M = 2000): head + tail backward in their 32-row forms 43.9 us,
# 64-row forms 49.8 us, linked 40.4 us.  In the joint step the alternative is the 64-row pair on HALF the CUs, and which wins depends
# on who ends the step: while the LLM chain ended with the Flow chains the linked form LOST (20.67 / 20.67 -> 20.90 / 20.89 ms); since the
# rel-pos dQ role and the delta hand-over took ~1.2 ms off the LLM chain's end (tools: CVFT_CHAIN_EVENTS) the Flow chains end the step
# and their latency counts: 20.38 / 20.39 / 20.31 -> 20.30 / 20.18 / 20.22 ms (three same-box pairs).  flow_only: 14.55 -> 14.47.
BLOCK_LINK_BWD = _os.environ.get("CVFT_BLOCK_LINK_BWD", "auto")


def block_link_bwd_on() -> bool:
    return BLOCK_LINK_BWD != "0"


BLOCK_QKV_FUSE = _os.environ.get("CVFT_BLOCK_QKV_FUSE", "1") != "0"      # first half of the block (norm1 + stacked LoRA q|k|v)
# its 64-rows-per-workgroup form (csrc/block_qkv_wide.hip): "0" off, "fwd" forward only (the two directions exchange only standard tensors),
# "1" / "both" forward and backward, "auto" (default) = both while >= 3 chains share the chip.  Same-box A/B, three pairs of 60 steps:
# joint 22.78 / 22.81 / 22.82 (off) against 22.59 / 22.58 / 22.45 (both); flow_only 14.7 -> 15.2 (fwd) / 17.2 (both, with the 64-row tail)
BLOCK_QKV_WIDE = _os.environ.get("CVFT_BLOCK_QKV_WIDE", "auto")


def block_qkv_wide() -> str:
    if BLOCK_QKV_WIDE != "auto":
        return BLOCK_QKV_WIDE
    return "both" if lib().cvft_concurrent_chains() >= 3 else "0"


# ---------------------------------------------------------------------------------
# 1-D convolutions as tap-GEMMs (frozen weights, no LoRA)
# ---------------------------------------------------------------------------------
class ConvPack:
    """Conv1d(k in {1,3}, stride in {1,2}, pad (k-1)/2) or ConvTranspose1d(k4,s2,p1), channel-last.
    Holds tap-major packed weights for forward and for the input-gradient pass."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], dtype: torch.dtype, stride: int = 1,
                 transposed: bool = False):
        w = weight.detach().float()
        self.transposed, self.stride = transposed, stride
        self.bias = None if bias is None else bias.detach().float().contiguous()
        if not transposed:
            self.N, self.Cin, self.k = w.shape                     # [Cout][Cin][k]
            assert self.k in (1, 3) and stride in (1, 2)
            self.Wf = w.permute(0, 2, 1).reshape(self.N, self.k * self.Cin).to(dtype).contiguous()
            if stride == 1:
                # dx[t] = sum_j dy[t + j - pad] . w[:, :, k-1-j]
                self.Wb = [w.flip(2).permute(1, 2, 0).reshape(self.Cin, self.k * self.N).to(dtype).contiguous()]
            else:
                assert self.k == 3
                # even rows 2m: dy[m].w1 ; odd rows 2m+1: dy[m].w2 + dy[m+1].w0
                self.Wb = [w[:, :, 1].t().to(dtype).contiguous(),
                           torch.cat([w[:, :, 2].t(), w[:, :, 0].t()], dim=1).to(dtype).contiguous()]
        else:
            self.Cin, self.N, self.k = w.shape                     # [Cin][Cout][4]
            assert self.k == 4 and stride == 2
            # even out 2m: x[m].w1 + x[m-1].w3 ; odd out 2m+1: x[m+1].w0 + x[m].w2
            self.Wf = [torch.cat([w[:, :, 1].t(), w[:, :, 3].t()], dim=1).to(dtype).contiguous(),
                       torch.cat([w[:, :, 0].t(), w[:, :, 2].t()], dim=1).to(dtype).contiguous()]
            # dx[t] = sum_j dy[2t - 1 + j] . w[:, :, j]^T
            self.Wb = [w.permute(0, 2, 1).reshape(self.Cin, 4 * self.N).to(dtype).contiguous()]

    def out_len(self, Tin: int) -> int:
        if self.transposed:
            return 2 * Tin
        return Tin if self.stride == 1 else (Tin - 1) // 2 + 1


# Two stride-1 convolutions reading the SAME x (ResnetBlock1D: block1's conv and res_conv, matcha decoder.py:76-94): autograd would
# add their two input gradients with a launch of its own (14 per Flow chain and step).  The later conv in forward order ("park")
# runs FIRST in backward: it parks its dx and reports a zero gradient; the earlier one ("take") adds the parked tensor in its
# dgrad epilogue.  The pair is armed in forward only when both saw the same buffer.
# The pair is matched in forward on the x OBJECT both convolutions receive (_H_FORK: the "take" conv's forward leaves its token on x,
# the "park" conv's forward takes it); in backward x may be gone, so the parked gradient travels under that token.
_PRE_DX = {}        # token -> dx parked by the "park" conv's backward
_FORK_TOKEN = [0]
CONV_FORK = _os.environ.get("CVFT_CONV_FORK", "1") != "0"


class ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, pack: ConvPack, B: int, Tin: int, Tout: int, in_len, out_len, fork=None):
        x = _c(x)
        ctx.fork, ctx.token = None, None
        if CONV_FORK and fork is not None and not pack.transposed and pack.stride == 1 and ctx.needs_input_grad[0]:
            if fork == "take":
                _FORK_TOKEN[0] += 1
                ctx.fork, ctx.token = "take", _FORK_TOKEN[0]
                _hand(x, _H_FORK, ctx.token)
            elif fork == "park":
                ctx.token = _take_hand(x, _H_FORK)
                if ctx.token is not None:
                    ctx.fork = "park"
        pad = (pack.k - 1) // 2
        if not pack.transposed:
            taps = tuple(range(-pad, pad + 1))
            geo = Geo(Tm=Tout, Tin=Tin, Tout=Tout, in_stride=pack.stride, taps=taps)
            y = gemm(x, pack.Wf, bias=pack.bias, geo=geo, nb=B, in_len=in_len, out_len=out_len,
                     residual=None if residual is None else _c(residual))
        else:
            assert residual is None
            y = torch.empty((B * Tout, pack.N), dtype=x.dtype, device=x.device)
            gemm(x, pack.Wf[0], bias=pack.bias, geo=Geo(Tm=Tin, Tin=Tin, Tout=Tout, out_stride=2, out_off=0, taps=(0, -1)),
                 nb=B, in_len=in_len, out_len=out_len, out=y)
            gemm(x, pack.Wf[1], bias=pack.bias, geo=Geo(Tm=Tin, Tin=Tin, Tout=Tout, out_stride=2, out_off=1, taps=(1, 0)),
                 nb=B, in_len=in_len, out_len=out_len, out=y)
        ctx.pack, ctx.B, ctx.Tin, ctx.Tout = pack, B, Tin, Tout
        ctx.in_len, ctx.out_len = in_len, out_len
        return y

    @staticmethod
    def backward(ctx, dy):
        pack, B, Tin, Tout = ctx.pack, ctx.B, ctx.Tin, ctx.Tout
        dy = _c(dy)
        if not ctx.needs_input_grad[0]:
            return None, (dy if ctx.needs_input_grad[1] else None), None, None, None, None, None, None, None
        # forward's output mask zeroes dy rows >= out_len; forward's input mask zeroes dx rows >= in_len
        kw = dict(nb=B, in_len=ctx.out_len, out_len=ctx.in_len)
        if not pack.transposed and pack.stride == 1:
            pad = (pack.k - 1) // 2
            # (the parked tensor carries the same row mask, so mask(dgrad + parked) == mask(dgrad) + parked)
            parked = _PRE_DX.pop(ctx.token, None) if ctx.fork == "take" else None
            dx = gemm(dy, pack.Wb[0], geo=Geo(Tm=Tin, Tin=Tout, Tout=Tin, taps=tuple(range(-pad, pad + 1))), residual=parked, **kw)
            if ctx.fork == "park":
                _PRE_DX[ctx.token] = dx
                return None, (dy if ctx.needs_input_grad[1] else None), None, None, None, None, None, None, None
        elif not pack.transposed:
            dx = torch.empty((B * Tin, pack.Cin), dtype=dy.dtype, device=dy.device)
            gemm(dy, pack.Wb[0], geo=Geo(Tm=(Tin + 1) // 2, Tin=Tout, Tout=Tin, out_stride=2, out_off=0, taps=(0,)), out=dx, **kw)
            if Tin // 2 > 0:
                gemm(dy, pack.Wb[1], geo=Geo(Tm=Tin // 2, Tin=Tout, Tout=Tin, out_stride=2, out_off=1, taps=(0, 1)), out=dx, **kw)
        else:
            dx = gemm(dy, pack.Wb[0], geo=Geo(Tm=Tin, Tin=Tout, Tout=Tin, in_stride=2, taps=(-1, 0, 1, 2)), **kw)
        # residual is added after the output mask only when out_len is None (the only use: ResnetBlock1D)
        dres = dy if ctx.needs_input_grad[1] else None
        return dx, dres, None, None, None, None, None, None, None


def conv1d(x, pack: ConvPack, B: int, Tin: int, Tout: Optional[int] = None, in_len=None, out_len=None, residual=None, fork=None):
    """x [B*Tin, Cin] -> [B*Tout, Cout]; in_len masks input frames (x*mask), out_len zeroes output frames;
    `residual` [B*Tout, Cout] is added in the GEMM epilogue (stride-1 convs, no out_len).
    fork: "take" on the first and "park" on the second of two stride-1 convs of the same x, the second one's output depending on the
    first's (so its backward runs first): their input gradients are summed inside the first conv's dgrad launch."""
    Tout = pack.out_len(Tin) if Tout is None else Tout
    assert residual is None or out_len is None
    return ConvFn.apply(x, residual, pack, B, Tin, Tout, in_len, out_len, fork)


# ---------------------------------------------------------------------------------
# normalisation
# ---------------------------------------------------------------------------------
class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps: float, relu: bool, post_scale: float):
        x = _c(x)
        rows, Cn = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        check(lib().cvft_layernorm_fwd(dt(x), rows, Cn, ptr(x), ptr(gamma), ptr(beta), eps, int(relu), post_scale,
                                       ptr(y), ptr(mean), ptr(rstd), stream()), "cvft_layernorm_fwd")
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.relu, ctx.post = relu, post_scale
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(x)
        check(lib().cvft_layernorm_bwd(dt(x), x.shape[0], x.shape[1], ptr(x), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd),
                                       int(ctx.relu), ctx.post, ptr(dy), None, ptr(dx), stream()), "cvft_layernorm_bwd")
        return dx, None, None, None, None, None


def layernorm(x, gamma, beta, eps: float = 1e-5, relu: bool = False, post_scale: float = 1.0):
    return LayerNormFn.apply(x, gamma, beta, eps, relu, post_scale)


class LayerNormForkFn(torch.autograd.Function):
    """(x, LN(x)) for the pre-norm residual pattern  x + f(LN(x)):  both gradient branches of x arrive at this one
    node, so  dx = d_residual + LN'(d_normed)  is ONE kernel (cvft_layernorm_bwd `dres`) instead of a LayerNorm
    backward plus an autograd accumulation add."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps: float, side=None):
        ctx.prev_odrop = _take_hand(x, _H_ODROP)                 # x = residual + dropout(linear(.)): (p, site) of that mask
        x = _c(x)
        rows, Cn = x.shape
        if side is not None:                  # (A [R, K] compute dtype, alpha, p, nsites): the adapter that will read y
            y, mean, rstd = ln_skinny_dropout(x, gamma, beta, eps, *side)
        else:
            y = torch.empty_like(x)
            mean = torch.empty(rows, dtype=torch.float32, device=x.device)
            rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
            check(lib().cvft_layernorm_fwd(dt(x), rows, Cn, ptr(x), ptr(gamma), ptr(beta), eps, 0, 1.0,
                                           ptr(y), ptr(mean), ptr(rstd), stream()), "cvft_layernorm_fwd")
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, dres, dy):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        if dy is None:
            return dres, None, None, None, None
        dy = _c(dy)
        dres = None if dres is None else _c(dres)
        dx = torch.empty_like(x)
        od = ctx.prev_odrop
        vec = 16 // x.element_size()
        if (od is not None and x.shape[1] % vec == 0 and x.shape[1] <= 256 * vec and
                all(t is None or (t.data_ptr() & 15) == 0 for t in (x, dy, dres, dx))):
            dxm = torch.empty_like(x)          # dx under the mask of x's producer, parked for that Function's backward
            side = od[2] if len(od) > 2 else None
            if side is not None and x.dtype == torch.bfloat16:
                Bt, sc = side
                V = torch.empty((x.shape[0], 16), dtype=x.dtype, device=x.device)
                check(lib().cvft_layernorm_bwd_mask_side(x.shape[0], x.shape[1], ptr(x), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd),
                                                         ptr(dy), ptr(dres), ptr(dx), od[0], ptr(_DROPOUT["seed"]), od[1], ptr(dxm),
                                                         ptr(Bt), 16, float(sc), ptr(V), stream()), "cvft_layernorm_bwd_mask_side")
                _hand(dx, _H_PRE_MASKED, (dxm, od[0], od[1], (V, Bt.data_ptr(), float(sc))))
                ctx.side_keep = Bt             # (Bt's address identifies the adapter: it stays this tensor's until the adapter's backward has asked)
                return dx, None, None, None, None
            check(lib().cvft_layernorm_bwd_mask(dt(x), x.shape[0], x.shape[1], ptr(x), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd),
                                                ptr(dy), ptr(dres), ptr(dx), od[0], ptr(_DROPOUT["seed"]), od[1], ptr(dxm), stream()),
                  "cvft_layernorm_bwd_mask")
            _hand(dx, _H_PRE_MASKED, (dxm, od[0], od[1], None))
            return dx, None, None, None, None
        check(lib().cvft_layernorm_bwd(dt(x), x.shape[0], x.shape[1], ptr(x), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd),
                                       0, 1.0, ptr(dy), ptr(dres), ptr(dx), stream()), "cvft_layernorm_bwd")
        return dx, None, None, None, None


def layernorm_fork(x, gamma, beta, eps: float = 1e-5, side=None):
    """-> (x_residual, LN(x)); use x_residual (not x) for the residual connection.
    side = (A [R, K], alpha, p, nsites): y feeds a LoRA adapter under lora_dropout -- its rank-side product comes out of
    the same launch (ln_skinny_dropout) and rides on y (_H_PRE_U) to that adapter's Function."""
    return LayerNormForkFn.apply(x, gamma, beta, eps, side)


class GroupNormMishFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, B: int, T: int, G: int, eps: float, length, add, mish: bool, t_eff=None):
        x = _c(x)
        Cn = x.shape[1]
        assert x.shape[0] == B * T
        y = torch.empty_like(x)
        mean = torch.empty(B * G, dtype=torch.float32, device=x.device)
        rstd = torch.empty(B * G, dtype=torch.float32, device=x.device)
        check(lib().cvft_groupnorm_mish_fwd(dt(x), B, T, Cn, G, ptr(x), ptr(gamma), ptr(beta), eps, ptr(length),
                                            ptr(None if add is None else _c(add)), int(mish), ptr(y), ptr(mean),
                                            ptr(rstd), ptr(t_eff), stream()), "cvft_groupnorm_mish_fwd")
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.dims, ctx.length, ctx.mish, ctx.t_eff = (B, T, Cn, G), length, mish, t_eff
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        B, T, Cn, G = ctx.dims
        dy = _c(dy)
        dx = torch.empty_like(x)
        ws = torch.empty(B * G * 2 * cb.GN_SPLIT, dtype=torch.float32, device=x.device)
        check(lib().cvft_groupnorm_mish_bwd(dt(x), B, T, Cn, G, ptr(x), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd),
                                            ptr(ctx.length), int(ctx.mish), ptr(dy), ptr(dx), ptr(ws), ptr(ctx.t_eff), stream()),
              "cvft_groupnorm_mish_bwd")
        return dx, None, None, None, None, None, None, None, None, None, None


def groupnorm_mish(x, gamma, beta, B: int, T: int, G: int, eps: float = 1e-5, length=None, add=None, mish: bool = True,
                   t_eff=None):
    """x [B*T, C] channel-last; y = mish(GN(x)) * (t < length[b]) + add[b]  (add carries no gradient).
    t_eff (device int32[1], optional): the exact batch's frame count when T is padded to a shape bucket (cvft.h)."""
    return GroupNormMishFn.apply(x, gamma, beta, B, T, G, eps, length, add, mish, t_eff)


# ---------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------
# The bf16 attention forward also writes O's rounding residual (include/cvft.h `o_lo`) when a backward will follow: the backward's
# delta = rowsum(dO (O + O_lo)) then carries fp32-level accuracy instead of bf16's (csrc/attn_common.h).  CVFT_ATTN_OLO=0: off.
ATTN_OLO = int(_os.environ.get("CVFT_ATTN_OLO", "1"))      # 0 off, 1 estimator (additive-bias) attention only, 2 rel-pos attention too


def _attn_residual(ctx, o: torch.Tensor, rel: bool = False):
    if ATTN_OLO >= (2 if rel else 1) and o.dtype == torch.bfloat16 and any(ctx.needs_input_grad[:3]):
        return torch.empty_like(o)
    return None


class AttnBiasFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, B: int, H: int, T: int, klen, scale: float, iso_len: int = 0):
        assert q.stride(1) == 1 and k.stride(1) == 1 and v.stride(1) == 1
        assert q.stride(0) == k.stride(0) == v.stride(0)
        o = torch.empty((B * T, H * 64), dtype=q.dtype, device=q.device)
        lse = torch.empty((B, H, T), dtype=torch.float32, device=q.device)
        o_lo = _attn_residual(ctx, o)
        with _Bracket("attn_bias_fwd", 4.0 * B * H * T * T * 64, 4.0 * B * T * H * 64 * q.element_size()):     # QK^T, PV
            check(lib().cvft_attn_bias_fwd(dt(q), B, H, T, ptr(q), ptr(k), ptr(v), q.stride(0), ptr(klen), scale, int(iso_len), ptr(o),
                                           o.stride(0), ptr(lse), ptr(o_lo), stream()), "cvft_attn_bias_fwd")
        ctx.save_for_backward(q, k, v, o, lse, o_lo)
        ctx.args = (B, H, T, klen, scale, int(iso_len))
        if TAIL_DELTA and q.dtype == torch.bfloat16:
            _hand(o, _H_ATTN_O, (o_lo, B, H, T))          # for the block tail that consumes o: its backward can form delta (below)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse, o_lo = ctx.saved_tensors
        B, H, T, klen, scale, iso_len = ctx.args
        pre = _take_hand(do, _H_DELTA)
        do = _c(do)
        dqkv = torch.empty((B * T, 3 * H * 64), dtype=q.dtype, device=q.device)
        dq, dk, dv = dqkv[:, :H * 64], dqkv[:, H * 64:2 * H * 64], dqkv[:, 2 * H * 64:]
        if pre is not None and tuple(pre.shape) == (B, H, T) and q.dtype == torch.bfloat16:
            # delta = rowsum(do . (o + o_lo)) came with do from the launch that produced it (the block tail's backward): neither
            # backward role reads o (cvft_attn_bias_bwd with o == NULL)
            delta, po, plo = pre, None, None
        else:
            delta, po, plo = torch.empty((B, H, T), dtype=torch.float32, device=q.device), o, o_lo
        with _Bracket("attn_bias_bwd", 10.0 * B * H * T * T * 64, 8.0 * B * T * H * 64 * q.element_size()):    # S, dP, dV, dK, dQ
            check(lib().cvft_attn_bias_bwd(dt(q), B, H, T, ptr(q), ptr(k), ptr(v), q.stride(0), ptr(klen), scale, iso_len, ptr(po),
                                           ptr(do), o.stride(0), ptr(lse), ptr(plo), ptr(delta), ptr(dq), ptr(dk), ptr(dv),
                                           dqkv.stride(0), stream()), "cvft_attn_bias_bwd")
        return dq, dk, dv, None, None, None, None, None, None


def attn_bias(q, k, v, B: int, H: int, T: int, klen, scale: float, iso_len: int = 0):
    """iso_len > 0: prompt-isolation mask -- frames [0, iso_len) and [iso_len, T) attend only within their segment."""
    return AttnBiasFn.apply(q, k, v, B, H, T, klen, scale, iso_len)


class AttnRelPosFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, p, bias_u, bias_v, B: int, H: int, L: int, length, causal: bool, scale: float,
                drop_p: float = 0.0, drop_site: int = 0):
        assert q.stride(1) == 1 and q.stride(0) == k.stride(0) == v.stride(0)
        assert p.shape[0] == 2 * L - 1 and p.stride(1) == 1
        seed = _DROPOUT["seed"] if drop_p > 0 else None
        o = torch.empty((B * L, H * 64), dtype=q.dtype, device=q.device)
        lse = torch.empty((B, H, L), dtype=torch.float32, device=q.device)
        o_lo = _attn_residual(ctx, o, rel=True)
        vis = 0.5 if causal else 1.0                      # share of the (query, key) square a causal launch has to compute
        with _Bracket("attn_relpos_fwd", 6.0 * vis * B * H * L * L * 64, 4.0 * B * L * H * 64 * q.element_size()):   # QK^T, band, PV
            check(lib().cvft_attn_relpos_fwd(dt(q), B, H, L, ptr(q), ptr(k), ptr(v), q.stride(0), ptr(p), p.stride(0),
                                             ptr(bias_u), ptr(bias_v), ptr(length), int(causal), scale, ptr(o), o.stride(0),
                                             ptr(lse), ptr(o_lo), float(drop_p), ptr(seed), drop_site, stream()), "cvft_attn_relpos_fwd")
        ctx.save_for_backward(q, k, v, p, bias_u, bias_v, o, lse, o_lo)
        ctx.args = (B, H, L, length, causal, scale)
        ctx.drop = (float(drop_p), seed, drop_site)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, p, bu, bv, o, lse, o_lo = ctx.saved_tensors
        B, H, L, length, causal, scale = ctx.args
        do = _c(do)
        dpos = None
        if ctx.needs_input_grad[3]:       # LoRA on linear_pos (reference lora.py:155-166 default targets): fp32 accumulator
            dpos = torch.zeros((2 * L - 1, H * 64), dtype=torch.float32, device=q.device)
        dqkv = torch.empty((B * L, 3 * H * 64), dtype=q.dtype, device=q.device)
        dq, dk, dv = dqkv[:, :H * 64], dqkv[:, H * 64:2 * H * 64], dqkv[:, 2 * H * 64:]
        delta = torch.empty((B, H, L), dtype=torch.float32, device=q.device)
        vis = 0.5 if causal else 1.0
        with _Bracket("attn_relpos_bwd", 14.0 * vis * B * H * L * L * 64, 8.0 * B * L * H * 64 * q.element_size()):  # S, band, dP, dV, dK, dQ (k and p terms)
            check(lib().cvft_attn_relpos_bwd(dt(q), B, H, L, ptr(q), ptr(k), ptr(v), q.stride(0), ptr(p), p.stride(0),
                                             ptr(bu), ptr(bv), ptr(length), int(causal), scale, ptr(o), ptr(do), o.stride(0),
                                             ptr(lse), ptr(o_lo), ptr(delta), ptr(dq), ptr(dk), ptr(dv), dqkv.stride(0), ptr(dpos),
                                             ctx.drop[0], ptr(ctx.drop[1]), ctx.drop[2], stream()),
                  "cvft_attn_relpos_bwd")
        dp_out = None if dpos is None else dpos.to(p.dtype)
        return dq, dk, dv, dp_out, None, None, None, None, None, None, None, None, None, None


def attn_relpos(q, k, v, p, bias_u, bias_v, B: int, H: int, L: int, length, causal: bool, scale: float, dropout_p: float = 0.0):
    """dropout_p > 0: attention-probability dropout (attention.py:118), mask re-derived in backward (device-side seed)."""
    site = 0
    if dropout_p > 0:
        if _DROPOUT["seed"] is None:
            dropout_begin_step()
        _DROPOUT["site"] += 1
        site = _DROPOUT["site"]
    return AttnRelPosFn.apply(q, k, v, p, bias_u, bias_v, B, H, L, length, causal, scale, dropout_p, site)


# ---------------------------------------------------------------------------------
# gathers, interpolation
# ---------------------------------------------------------------------------------
def embed_gather(tok: torch.Tensor, table: torch.Tensor, length: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[B, L] int64 -> [B*L, D]; rows l >= length[b] are zero; negative ids clamp to 0. No gradient (frozen table)."""
    B, L = tok.shape
    out = torch.empty((B * L, table.shape[1]), dtype=table.dtype, device=table.device)
    check(lib().cvft_embed_gather(dt(table), B, L, table.shape[1], ptr(_c(tok)), ptr(length), ptr(table), ptr(out), stream()),
          "cvft_embed_gather")
    return out


class GatherRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, idx, fill: float):
        src = _c(src)
        n, D = idx.numel(), src.shape[1]
        out = torch.empty((n, D), dtype=src.dtype, device=src.device)
        check(lib().cvft_gather_rows(dt(src), n, D, ptr(idx), ptr(src), fill, ptr(out), stream()), "cvft_gather_rows")
        ctx.save_for_backward(idx)
        ctx.src_rows = src.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        dout = _c(dout)
        dsrc = torch.zeros((ctx.src_rows, dout.shape[1]), dtype=dout.dtype, device=dout.device)
        check(lib().cvft_scatter_rows(dt(dout), idx.numel(), dout.shape[1], ptr(idx), ptr(dout), ptr(dsrc), stream()),
              "cvft_scatter_rows")
        return dsrc, None, None


def gather_rows(src, idx: torch.Tensor, fill: float = 0.0):
    """out[i] = src[idx[i]] (idx >= 0) else `fill`; each source row may be referenced at most once."""
    return GatherRowsFn.apply(src, idx, fill)


class InterpLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, B: int, Lin: int, Lout: int, eff=None):
        x = _c(x)
        Cn = x.shape[1]
        y = torch.empty((B * Lout, Cn), dtype=x.dtype, device=x.device)
        check(lib().cvft_interp_linear_fwd(dt(x), B, Lin, Lout, Cn, ptr(x), ptr(y), ptr(eff), stream()), "cvft_interp_linear_fwd")
        ctx.dims, ctx.eff = (B, Lin, Lout, Cn), eff
        return y

    @staticmethod
    def backward(ctx, dy):
        B, Lin, Lout, Cn = ctx.dims
        dy = _c(dy)
        dx = torch.empty((B * Lin, Cn), dtype=dy.dtype, device=dy.device)
        check(lib().cvft_interp_linear_bwd(dt(dy), B, Lin, Lout, Cn, ptr(dy), ptr(dx), ptr(ctx.eff), stream()), "cvft_interp_linear_bwd")
        return dx, None, None, None, None


def interp_linear(x, B: int, Lin: int, Lout: int, eff=None):
    """eff (device int32[2], optional): the exact batch's (Lin, Lout) when the tensors are padded to shape buckets (cvft.h)."""
    return InterpLinearFn.apply(x, B, Lin, Lout, eff)


def l2norm_rows(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    x = _c(x.float())
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    check(lib().cvft_l2norm_rows(cb.F32 if dtype == torch.float32 else cb.BF16, x.shape[0], x.shape[1], ptr(x), ptr(y),
                                 stream()), "cvft_l2norm_rows")
    return y


def time_embed(t: torch.Tensor, freqs: torch.Tensor, dtype: torch.dtype, scale: float = 1000.0) -> torch.Tensor:
    B, dim = t.numel(), 2 * freqs.numel()
    out = torch.empty((B, dim), dtype=dtype, device=t.device)
    check(lib().cvft_time_embed(cb.F32 if dtype == torch.float32 else cb.BF16, B, dim, ptr(_c(t.float())), ptr(freqs),
                                scale, ptr(out), stream()), "cvft_time_embed")
    return out


# ---------------------------------------------------------------------------------
# CFM prepare + losses
# ---------------------------------------------------------------------------------
class CfmPrepareFn(torch.autograd.Function):
    """feat/z/t_raw/keep -> packed estimator input [B*T, 320] = [y | mu*keep | spk*keep | 0], target u, t.
    Gradient flows to mu only (dmu = dxin[:, 80:160] * keep)."""

    @staticmethod
    def forward(ctx, mu, spk, feat, z, t_raw, keep, B: int, T: int, mel_mean: float, mel_std: float, sigma_min: float,
                cond=None, cosine: bool = True):
        mu = _c(mu)
        xin = torch.empty((B * T, 320), dtype=mu.dtype, device=mu.device)
        u = torch.empty((B * T, 80), dtype=torch.float32, device=mu.device)
        t = torch.empty(B, dtype=torch.float32, device=mu.device)
        check(lib().cvft_cfm_prepare(dt(mu), B, T, ptr(_c(feat)), ptr(_c(z)), ptr(_c(t_raw)), ptr(_c(keep)), ptr(mu),
                                     ptr(_c(spk)), ptr(None if cond is None else _c(cond)), mel_mean, mel_std, sigma_min, int(cosine), ptr(xin), ptr(u), ptr(t), stream()),
              "cvft_cfm_prepare")
        ctx.save_for_backward(keep)
        ctx.dims = (B, T)
        ctx.mark_non_differentiable(u, t)
        return xin, u, t

    @staticmethod
    def backward(ctx, dxin, du, dt_):
        (keep,) = ctx.saved_tensors
        B, T = ctx.dims
        dmu = (dxin[:, 80:160].reshape(B, T, 80) * keep.view(B, 1, 1).to(dxin.dtype)).reshape(B * T, 80)
        return dmu, None, None, None, None, None, None, None, None, None, None, None, None


def cfm_prepare(mu, spk, feat, z, t_raw, keep, B: int, T: int, mel_mean: float, mel_std: float, sigma_min: float,
                cond=None, cosine: bool = True):
    """cosine: the reference's t_scheduler == 'cosine' (flow_matching.py:176); False leaves t as drawn."""
    return CfmPrepareFn.apply(mu, spk, feat, z, t_raw, keep, B, T, mel_mean, mel_std, sigma_min, cond, cosine)


class MaskedMseFn(torch.autograd.Function):
    """sum(((pred-u)*mask)^2) / denom   (reference flow_matching.py:192), denom a device scalar."""

    @staticmethod
    def forward(ctx, pred, u, length, denom, B: int, T: int, weight=None):
        pred = _c(pred)
        Cn = pred.shape[1]
        s = torch.zeros(1, dtype=torch.float32, device=pred.device)
        ctx.weight = None if weight is None else _c(weight.float())
        check(lib().cvft_masked_mse_fwd(dt(pred), B, T, Cn, ptr(pred), ptr(u), ptr(length), ptr(ctx.weight), ptr(s), stream()),
              "cvft_masked_mse_fwd")
        ctx.save_for_backward(pred, u, length, denom)
        ctx.dims = (B, T, Cn)
        return (s / denom).squeeze(0)

    @staticmethod
    def backward(ctx, g):
        pred, u, length, denom = ctx.saved_tensors
        B, T, Cn = ctx.dims
        gs = (g.float() / denom).reshape(1).contiguous()
        dpred = torch.empty_like(pred)
        check(lib().cvft_masked_mse_bwd(dt(pred), B, T, Cn, ptr(pred), ptr(u), ptr(length), ptr(ctx.weight), ptr(gs), ptr(dpred),
                                        stream()), "cvft_masked_mse_bwd")
        return dpred, None, None, None, None, None, None


def masked_mse(pred, u, length, denom, B: int, T: int, weight=None):
    """weight [B*T] fp32 (optional): per-frame loss weights, entering squared like the reference's (pred-u)*loss_mask."""
    return MaskedMseFn.apply(pred, u, length, denom, B, T, weight)


class CrossEntropyFn(torch.autograd.Function):
    """Token-mean label-smoothed CE with ignore index -1 (label_smoothing_loss.py:68-96) + accuracy."""

    @staticmethod
    def forward(ctx, logits, target, smoothing: float = 0.0):
        logits = _rowc(logits)
        ctx.smoothing = float(smoothing)
        n, V = logits.shape
        out3 = torch.zeros(3, dtype=torch.float32, device=logits.device)
        row_lse = torch.empty(n, dtype=torch.float32, device=logits.device)
        check(lib().cvft_ce_fwd(dt(logits), n, V, ptr(logits), logits.stride(0), ptr(target), ptr(out3), ptr(row_lse),
                                ctx.smoothing, stream()), "cvft_ce_fwd")
        ctx.save_for_backward(logits, target, row_lse, out3)
        q = out3 / out3[1]                    # one launch for both quotients (this sits between the LLM's forward and backward)
        loss, acc = q[0], q[2]
        ctx.mark_non_differentiable(acc)
        return loss, acc

    @staticmethod
    def backward(ctx, g, gacc):
        logits, target, row_lse, out3 = ctx.saved_tensors
        n, V = logits.shape
        gs = (g.float() / out3[1]).reshape(1).contiguous()
        pitch = logits.stride(0)
        base = torch.empty((n, pitch), dtype=logits.dtype, device=logits.device)   # same row pitch as the logits
        dl = base[:, :V]
        if pitch > V:
            base[:, V:].zero_()
            _register_zero_padded(base)
        check(lib().cvft_ce_bwd(dt(logits), n, V, ptr(logits), logits.stride(0), ptr(target), ptr(row_lse), ptr(gs),
                                ptr(dl), dl.stride(0), ctx.smoothing, stream()), "cvft_ce_bwd")
        return dl, None, None


def cross_entropy(logits, target, smoothing: float = 0.0):
    """logits [n, V]; target [n] int32 (-1 = ignore) -> (loss, accuracy).  smoothing = LabelSmoothingLoss's lsm_weight."""
    return CrossEntropyFn.apply(logits, target, smoothing)


# ---------------------------------------------------------------------------------
# depthwise conv (Conformer ConvolutionModule; not executed by the 300M config)
# ---------------------------------------------------------------------------------
class DwConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, B: int, T: int, pad_left: int, length):
        x = _c(x)
        Cn, Kw = w.shape
        y = torch.empty_like(x)
        check(lib().cvft_dwconv1d_fwd(dt(x), B, T, Cn, Kw, pad_left, ptr(x), ptr(w), ptr(bias), ptr(length), ptr(y), stream()),
              "cvft_dwconv1d_fwd")
        ctx.save_for_backward(w)
        ctx.args = (B, T, pad_left, length)
        return y

    @staticmethod
    def backward(ctx, dy):
        (w,) = ctx.saved_tensors
        B, T, pad_left, length = ctx.args
        dy = _c(dy)
        dx = torch.empty_like(dy)
        check(lib().cvft_dwconv1d_bwd(dt(dy), B, T, w.shape[0], w.shape[1], pad_left, ptr(dy), ptr(w), ptr(length), ptr(dx),
                                      stream()), "cvft_dwconv1d_bwd")
        return dx, None, None, None, None, None, None


def dwconv1d(x, w, bias, B: int, T: int, pad_left: int, length=None):
    """x [B*T, C] channel-last, w [C, Kw] fp32; input frames t >= length[b] are treated as zero."""
    return DwConvFn.apply(x, w, bias, B, T, pad_left, length)
