"""HIP-backed building blocks with the reference's module tree / state-dict keys.

Leaf modules are ordinary ``nn.Linear`` / ``nn.Conv1d`` / ``nn.LayerNorm`` / ``nn.GroupNorm`` /
``nn.Embedding`` objects (so ``state_dict()`` keys, ``lora.apply_lora_to_model`` name matching
and ``load_state_dict(strict=True)`` behave exactly like the reference's modules.py /
cosyvoice.transformer.*), but they are used as *parameter containers only*: every forward in
this file runs hand-written gfx950 kernels through ``hipops.functional`` on channel-last
``[batch*time, channels]`` activations.  Lengths travel as int32 device tensors -- no
(B,T,T) mask / bias tensors and no host syncs inside a step.

Reference mapping: modules.py:20-1106 (self-contained twin) == cosyvoice/flow/decoder.py,
matcha/models/components/{decoder,transformer}.py (vendored); cosyvoice/transformer/
{encoder,encoder_layer,attention,embedding,subsampling,positionwise_feed_forward}.py.
"""
from __future__ import annotations

import math
import os
import warnings
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from .hipops import functional as HF
from .lora import LoRAConv1d, LoRALinear


@dataclass
class Numerics:
    """Numerics switches.  Defaults = the vendored path train_joint.py runs."""
    dtype: torch.dtype = torch.float32     # activation / frozen-weight storage + MFMA input type
    gelu: str = "gelu_erf"                 # diffusers GELU (erf); twin modules.GELU uses "gelu_tanh"
    xscale: bool = True                    # x*sqrt(d) in EspnetRelPositionalEncoding (twin omits it)
    enc_ln_eps: float = 1e-12              # encoder-layer LayerNorm eps (twin: 1e-5)

    @staticmethod
    def twin(dtype=torch.float32) -> "Numerics":
        return Numerics(dtype=dtype, gelu="gelu_tanh", xscale=False, enc_ln_eps=1e-5)


# ---------------------------------------------------------------------------------
# weight packs cached on the parameter-container modules
# ---------------------------------------------------------------------------------
def _cached(mod: nn.Module, key: str, weight: torch.Tensor, dtype, build):
    tag = (dtype, weight._version, weight.data_ptr())
    cache = mod.__dict__.setdefault("_cvft_cache", {})
    hit = cache.get(key)
    if hit is None or hit[0] != tag:
        hit = (tag, build())
        cache[key] = hit
    return hit[1]


def _lin_parts(mod: nn.Module):
    """(weight 2-D, bias, A, B, scale, dropout) of nn.Linear / LoRALinear / 1x1 Conv1d / LoRAConv1d."""
    A = Bm = drop = None
    scale = 1.0
    base = mod
    if isinstance(mod, LoRALinear):
        base, A, Bm, scale, drop = mod.original_layer, mod.lora_A, mod.lora_B, mod.scaling, mod.lora_dropout
    elif isinstance(mod, LoRAConv1d):
        base, scale, drop = mod.original_layer, mod.scaling, mod.lora_dropout
        A, Bm = mod.lora_A.weight.squeeze(-1), mod.lora_B.weight.squeeze(-1)
    w = base.weight
    if w.dim() == 3:
        assert w.shape[-1] == 1
        w = w.squeeze(-1)
    return base, w, base.bias, A, Bm, scale, drop


def hip_linear(mod: nn.Module, x: torch.Tensor, dtype: Optional[torch.dtype] = None, act: Optional[str] = None,
               residual: Optional[torch.Tensor] = None, out_drop: float = 0.0) -> torch.Tensor:
    """x [rows, in] -> [rows, out] through the fused LoRA tap-GEMM.  `mod` is an nn.Linear,
    a 1x1 nn.Conv1d, or their LoRA wrappers.  out_drop > 0: residual + dropout(linear(x)) (encoder_layer.py:95 / 104) with
    the mask applied in the GEMM epilogue (HF.OUT_DROP_FUSE, out features % 4 == 0), else a cvft_dropout_add pass."""
    dtype = x.dtype if dtype is None else dtype
    base, w, b, A, Bm, scale, drop = _lin_parts(mod)
    pack = _cached(base, "lin", base.weight, dtype, lambda: HF.LinearPack(w, b, dtype))
    if x.dtype != dtype:
        x = x.to(dtype)
    od = out_drop if (out_drop > 0 and HF.OUT_DROP_FUSE and pack.N % 4 == 0) else 0.0
    if od == 0.0 and out_drop > 0:
        return HF.dropout_add(hip_linear(mod, x, dtype, act), out_drop, residual)
    if A is not None and _lora_dropout_on(mod, drop):
        # reference lora.py:70: dropout on the side-path input only.  Fused form (mask inside the rank-side kernels,
        # counter-based) when the shapes allow, else nn.Dropout + separate side-path launches
        if type(drop) is nn.Dropout and HF._can_drop_fuse(x, A.shape[0]):
            y = HF.lora_linear(x, pack, A, Bm, scale, act, residual, drop_p=drop.p, out_drop_p=od)
            HF.drop_pre_u(x)                                 # (a hand-off from the LayerNorm launch nobody took is dropped here)
            return y
        if od > 0:
            return HF.dropout_add(_lora_dropout_path(x, pack, A, Bm, scale, drop, act, None), od, residual)
        return _lora_dropout_path(x, pack, A, Bm, scale, drop, act, residual)
    return HF.lora_linear(x, pack, A, Bm, scale, act, residual, out_drop_p=od)


def _lora_dropout_on(mod: nn.Module, drop) -> bool:
    return mod.training and isinstance(drop, nn.Dropout) and drop.p > 0


def hip_qkv(mq: nn.Module, mk: nn.Module, mv: nn.Module, x: torch.Tensor):
    """(q, k, v) projections of one input with the input-gradient accumulation fused (HF.LinearQKVFn)."""
    parts = [_lin_parts(m) for m in (mq, mk, mv)]
    drops = [p[6].p if (p[3] is not None and _lora_dropout_on(m, p[6])) else 0.0 for m, p in zip((mq, mk, mv), parts)]
    packs = [_cached(p[0], "lin", p[0].weight, x.dtype, lambda p=p: HF.LinearPack(p[1], p[2], x.dtype)) for p in parts]
    if any(d > 0 for d in drops):
        out = None
        if drops[0] == drops[1] == drops[2] and all(type(p[6]) is nn.Dropout for p in parts):
            out = HF.lora_linear_qkv(x, packs, [(p[3], p[4]) for p in parts], [p[5] for p in parts], drop_p=drops[0])
        HF.drop_pre_u(x)                                     # (a hand-off from the LayerNorm launch nobody took is dropped here)
        return out if out is not None else (hip_linear(mq, x), hip_linear(mk, x), hip_linear(mv, x))
    return HF.lora_linear_qkv(x, packs, [(p[3], p[4]) for p in parts], [p[5] for p in parts])


def hip_ffn(m1: nn.Module, m2: nn.Module, x: torch.Tensor, act: str, residual: Optional[torch.Tensor] = None,
            inner_drop: float = 0.0, out_drop: float = 0.0):
    """residual + drop_out(W2 drop_in(act(W1 x))) with the activation backward (and the inner mask) fused into W2's dgrad
    (HF.FeedForwardFn); inner_drop / out_drop: the encoder's training dropouts (masks in the GEMM epilogues)."""
    p1, p2 = _lin_parts(m1), _lin_parts(m2)
    if (inner_drop > 0 or out_drop > 0) and not (HF.FFN_TRAIN_FUSE and HF.OUT_DROP_FUSE and p1[1].shape[0] % 4 == 0 and p2[1].shape[0] % 4 == 0):
        h = hip_linear(m1, x)
        h = HF.act_dropout(h, act, inner_drop) if inner_drop > 0 else HF.ActFn.apply(h, act)
        return hip_linear(m2, h, residual=residual, out_drop=out_drop)
    d1 = p1[6].p if (p1[3] is not None and _lora_dropout_on(m1, p1[6])) else 0.0
    d2 = p2[6].p if (p2[3] is not None and _lora_dropout_on(m2, p2[6])) else 0.0
    if d1 > 0 or d2 > 0:
        ok = x.dtype == torch.bfloat16 and x.is_contiguous() and x.data_ptr() % 16 == 0 and \
            (d1 == 0 or (type(p1[6]) is nn.Dropout and p1[3].shape[0] == 16 and p1[1].shape[1] % 32 == 0)) and \
            (d2 == 0 or (type(p2[6]) is nn.Dropout and p2[3].shape[0] == 16 and p2[1].shape[1] % 32 == 0))
        if not ok:
            h = hip_linear(m1, x, act=None if inner_drop > 0 else act)
            if inner_drop > 0:
                h = HF.act_dropout(h, act, inner_drop)
            return hip_linear(m2, h, residual=residual, out_drop=out_drop)
    k1 = _cached(p1[0], "lin", p1[0].weight, x.dtype, lambda: HF.LinearPack(p1[1], p1[2], x.dtype))
    k2 = _cached(p2[0], "lin", p2[0].weight, x.dtype, lambda: HF.LinearPack(p2[1], p2[2], x.dtype))
    y = HF.lora_feed_forward(x, k1, k2, (p1[3], p1[4]), (p2[3], p2[4]), p1[5], p2[5], act, residual, p1=d1, p2=d2,
                             p_in=inner_drop, p_out=out_drop)
    HF.drop_pre_u(x)                                         # (a hand-off from the LayerNorm launch nobody took is dropped here)
    return y


def _lora_dropout_path(x, pack, A, Bm, scale, drop, act, residual):
    """y = act(x W^T + b + s * (drop(x) A^T) B^T) (+ residual)  (lora.py:64-76 with lora_dropout active): main GEMM
    without side path, then the rank-r side path on the dropped input (HF.LoraSideFn), then the activation."""
    if act is None:
        y = HF.lora_linear(x, pack, None, None, 1.0, None, residual)
        return HF.lora_side(drop(x), A, Bm, y, scale)
    z = HF.lora_side(drop(x), A, Bm, HF.lora_linear(x, pack, None, None, 1.0, None, None), scale)
    h = HF.ActFn.apply(z, act)
    return h if residual is None else h + residual


def conv_pack(mod: nn.Module, dtype) -> HF.ConvPack:
    transposed = isinstance(mod, nn.ConvTranspose1d)
    return _cached(mod, "conv", mod.weight, dtype,
                   lambda: HF.ConvPack(mod.weight, mod.bias, dtype, stride=mod.stride[0], transposed=transposed))


def _f32(p: torch.Tensor) -> torch.Tensor:
    return p.detach() if p.dtype == torch.float32 else p.detach().float()


def hip_layernorm(ln: nn.LayerNorm, x, eps: Optional[float] = None, relu=False, post_scale=1.0):
    return HF.layernorm(x, _f32(ln.weight), _f32(ln.bias), ln.eps if eps is None else eps, relu, post_scale)


def hip_layernorm_fork(ln: nn.LayerNorm, x, eps: Optional[float] = None, consumers=None):
    """-> (x_residual, LN(x)) for pre-norm residual blocks: the two gradient branches of x meet in one kernel.
    consumers: the LoRA module(s) LN(x) goes to next -- (linear_q, linear_k, linear_v) or (w_1,).  When they are about to take
    the fused lora_dropout path (hip_qkv / hip_linear below), the LayerNorm launch also emits their rank-side product
    U = s drop(LN(x)) A^T and the dropped copies (cvft_ln_skinny_dropout): one launch less per adapter and step."""
    g, b, e = _f32(ln.weight), _f32(ln.bias), ln.eps if eps is None else eps
    side = _ln_side(x, g, b, consumers) if consumers is not None else None
    return HF.layernorm_fork(x, g, b, e, side)


def _ln_side(x, g, b, mods):
    """(A operand, scale, p, nsites) when `mods` will run HF.skinny_dropout on LN(x) with exactly these operands (the
    conditions of hip_qkv's stacked path / hip_linear's fused-dropout path), else None."""
    parts = [_lin_parts(m) for m in mods]
    if any(p[3] is None or not _lora_dropout_on(m, p[6]) or type(p[6]) is not nn.Dropout for m, p in zip(mods, parts)):
        return None
    if len(mods) == 3:
        if not (parts[0][6].p == parts[1][6].p == parts[2][6].p) or not HF._can_drop_fuse(x, 48):
            return None
        packs = [_cached(p[0], "lin", p[0].weight, x.dtype, lambda p=p: HF.LinearPack(p[1], p[2], x.dtype)) for p in parts]
        st = HF._qkv_stacked_operands(x, packs, [(p[3], p[4]) for p in parts], [p[5] for p in parts])
        if st is None:
            return None
        A, scale, nsites = st[1][0], parts[0][5], 3
    else:
        if not HF._can_drop_fuse(x, parts[0][3].shape[0]):
            return None
        A, scale, nsites = HF._lora_operands(parts[0][3], x.dtype)[0], parts[0][5], 1
    return (A, scale, parts[0][6].p, nsites) if HF.can_ln_skinny(x, A, g, b) else None


def to_len(lens: torch.Tensor, device) -> torch.Tensor:
    return lens.to(device=device, dtype=torch.int32).contiguous()


# =================================================================================
# U-Net1D estimator bricks (matcha decoder.py:14-158 == modules.py:20-120)
# =================================================================================
class SinusoidalPosEmb(nn.Module):
    """modules.py:20-42; the constant frequency table is built on the host with the reference's
    exact torch ops, the sin/cos of scale*t*f runs in the HIP time-embed kernel."""

    def __init__(self, dim):
        super().__init__()
        assert dim % 2 == 0
        self.dim = dim
        half = dim // 2
        self._freqs_cpu = torch.exp(torch.arange(half).float() * -(math.log(10000) / (half - 1)))
        self._freqs_dev = None

    def freqs(self, device):
        if self._freqs_dev is None or self._freqs_dev.device != device:
            self._freqs_dev = self._freqs_cpu.to(device)
        return self._freqs_dev

    def forward(self, t, scale=1000, dtype=torch.float32):
        if t.ndim < 1:
            t = t.unsqueeze(0)
        return HF.time_embed(t, self.freqs(t.device), dtype, float(scale))


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu"):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act_name = "silu" if act_fn == "silu" else "mish"
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample):
        return hip_linear(self.linear_2, hip_linear(self.linear_1, sample, act=self.act_name))


class Block1D(nn.Module):
    """Conv1d(k3,p1) + GroupNorm(groups) + Mish with mask before/after (modules.py:60-73).
    Channel-last: x [B*T, C]; `length` int32 [B]; optional per-(b,c) additive term fused after the mask."""

    def __init__(self, dim, dim_out, groups=8):
        super().__init__()
        self.block = nn.Sequential(nn.Conv1d(dim, dim_out, 3, padding=1), nn.GroupNorm(groups, dim_out), nn.Mish())

    def forward(self, x, B, T, length, add=None, t_eff=None, fork=None):
        conv, gn = self.block[0], self.block[1]
        h = HF.conv1d(x, conv_pack(conv, x.dtype), B, T, in_len=length, fork=fork)
        return HF.groupnorm_mish(h, _f32(gn.weight), _f32(gn.bias), B, T, gn.num_groups, gn.eps, length, add, True, t_eff)


class ResnetBlock1D(nn.Module):
    """modules.py:76-94."""

    def __init__(self, dim, dim_out, time_emb_dim, groups=8):
        super().__init__()
        self.mlp = nn.Sequential(nn.Mish(), nn.Linear(time_emb_dim, dim_out))
        self.block1 = Block1D(dim, dim_out, groups=groups)
        self.block2 = Block1D(dim_out, dim_out, groups=groups)
        self.res_conv = nn.Conv1d(dim, dim_out, 1)

    def forward(self, x, B, T, length, temb_mish, t_eff=None):
        with torch.no_grad():
            add = hip_linear(self.mlp[1], temb_mish)                       # [B, dim_out]; depends on t only
        # x feeds block1's conv and res_conv: the two input gradients meet inside block1's dgrad launch (HF.conv1d `fork`)
        h = self.block1(x, B, T, length, add=add, t_eff=t_eff, fork="take")
        h = self.block2(h, B, T, length, t_eff=t_eff)
        # h + res_conv(x * mask): 1x1 conv == tap-GEMM with the input-length mask, residual fused
        return HF.conv1d(x, conv_pack(self.res_conv, x.dtype), B, T, in_len=length, residual=h, fork="park")


class Downsample1D(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.conv = nn.Conv1d(dim, dim, 3, 2, 1)


class Upsample1D(nn.Module):
    def __init__(self, dim, use_conv_transpose=True):
        super().__init__()
        assert use_conv_transpose
        self.conv = nn.ConvTranspose1d(dim, dim, 4, 2, 1)


# ---- diffusers-style transformer block (modules.py:127-375) ----------------------
class GELU(nn.Module):
    def __init__(self, dim_in, dim_out, approximate="none"):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out)
        self.approximate = approximate


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="gelu"):
        super().__init__()
        if activation_fn not in ("gelu", "gelu-approximate"):
            raise NotImplementedError(f"activation_fn={activation_fn!r}: only the CosyVoice-300M 'gelu' estimator is built")
        inner = int(dim * mult)
        self.net = nn.ModuleList([GELU(dim, inner, "tanh" if activation_fn == "gelu-approximate" else "none"),
                                  nn.Dropout(dropout), nn.Linear(inner, dim_out or dim)])


class Attention(nn.Module):
    def __init__(self, query_dim, heads=8, dim_head=64, dropout=0.0, bias=False):
        super().__init__()
        if dim_head != 64:
            raise NotImplementedError("fused attention kernels are specialised to head_dim 64")
        inner = dim_head * heads
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(query_dim, inner, bias=bias)
        self.to_v = nn.Linear(query_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(dropout)])


class BasicTransformerBlock(nn.Module):
    """x += Attn(LN(x), key-padding bias) ; x += W2 GELU(W1 LN(x))   (modules.py:349-375)."""

    def __init__(self, dim, num_attention_heads, attention_head_dim, dropout=0.0, activation_fn="gelu"):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, num_attention_heads, attention_head_dim, dropout)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim, dropout=dropout, activation_fn=activation_fn)

    def forward(self, x, B, T, length, gelu: str, iso_len: int = 0, next_tb=None):
        """next_tb: the block that reads this block's output next (same stage), or None -- its norm1 + q|k|v head then rides in
        this block's tail launch (HF.block_tail(link=...))."""
        a = self.attn1
        head = self._head_fused(x)
        if head is not None:           # norm1 + stacked LoRA q|k|v projection as one row-tile chain launch each way
            x, q, k, v = HF.block_qkv(x, *head)
        else:
            x, y = hip_layernorm_fork(self.norm1, x, consumers=(a.to_q, a.to_k, a.to_v))
            q, k, v = hip_qkv(a.to_q, a.to_k, a.to_v, y)
        o = HF.attn_bias(q, k, v, B, a.heads, T, length, a.scale, iso_len)
        return self._tail(o, x, gelu, next_tb)

    def _head_fused(self, x):
        """(loras, pack, stacked operands, scale, p) for HF.block_qkv when the block's first half can take the row-tile chain
        kernels: bf16, d = 256, three rank-16 LoRALinear projections of 512 features with one lora_dropout rate, frozen base
        weights / norm1, and the optimiser-maintained stacked bf16 shadows (optim.FlatAdamW.stack_for)."""
        a, ln = self.attn1, self.norm1
        mods = (a.to_q, a.to_k, a.to_v)
        if not (HF.BLOCK_FUSE and HF.BLOCK_QKV_FUSE and x.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[1] == 256
                and all(isinstance(m, LoRALinear) for m in mods)):
            return None
        parts = [_lin_parts(m) for m in mods]
        if any(p[1].shape != (512, 256) or p[3].shape[0] != 16 or p[1].requires_grad or (p[2] is not None and p[2].requires_grad)
               for p in parts) or ln.weight.requires_grad or ln.bias.requires_grad:
            return None
        drops = [p[6].p if _lora_dropout_on(m, p[6]) else 0.0 for m, p in zip(mods, parts)]
        if not (drops[0] == drops[1] == drops[2]) or any(d > 0 and type(p[6]) is not nn.Dropout for d, p in zip(drops, parts)):
            return None
        packs = [_cached(p[0], "lin", p[0].weight, x.dtype, lambda p=p: HF.LinearPack(p[1], p[2], x.dtype)) for p in parts]
        loras = [(p[3], p[4]) for p in parts]
        st = HF._qkv_stacked_operands(x, packs, loras, [p[5] for p in parts])
        if st is None:
            return None
        ws, ops = st
        tag = (ws.Wf.data_ptr(), ws.Wf._version, ln.weight._version, ln.weight.data_ptr(), ln.bias._version)
        hit = self.__dict__.get("_cvft_head")
        if hit is None or hit[0] != tag:
            from .hipops.blockpack import BlockQkvPack
            hit = (tag, BlockQkvPack(ws.Wf, ws.bias, ln.weight, ln.bias, ln.eps))
            self.__dict__["_cvft_head"] = hit
        return tuple(loras), hit[1], ops, parts[0][5], drops[0]

    def _tail_pack(self):
        """Packed frozen weights of the block's second half for the row-tile chain kernels, or None when a layer of it carries
        an adapter / trains (the flow target list has none on to_out / ff.net.*: config.py FLOW LoRA target_modules)."""
        to_out, ln, w1, w2 = self.attn1.to_out[0], self.norm3, self.ff.net[0].proj, self.ff.net[2]
        if not all(type(m) is nn.Linear for m in (to_out, w1, w2)):
            return None
        ps = [to_out.weight, to_out.bias, ln.weight, ln.bias, w1.weight, w1.bias, w2.weight, w2.bias]
        if any(p is not None and p.requires_grad for p in ps) or self.ff.net[1].p > 0 or self.attn1.to_out[1].p > 0:
            return None
        tag = tuple((p._version, p.data_ptr()) for p in ps if p is not None)
        hit = self.__dict__.get("_cvft_tail")
        if hit is None or hit[0] != tag:
            from .hipops.blockpack import BlockTailPack
            hit = (tag, BlockTailPack(to_out.weight, to_out.bias, ln.weight, ln.bias, ln.eps, w1.weight, w1.bias, w2.weight, w2.bias))
            self.__dict__["_cvft_tail"] = hit
        return hit[1]

    def _link(self, pack, next_tb, x):
        """link argument of HF.block_tail: the next block's head operands + the linked weight stream, or None"""
        if next_tb is None or not HF.block_link_on() or pack.DI != 512 or not 256 <= pack.F <= 1024:
            return None
        head = next_tb._head_fused(x)
        if head is None:
            return None
        loras, hpack, ops, scale, p = head
        hit = self.__dict__.get("_cvft_link")
        if hit is None or hit[0] is not pack or hit[1] is not hpack:
            from .hipops.blockpack import BlockLinkPack
            hit = (pack, hpack, BlockLinkPack(pack, hpack))
            self.__dict__["_cvft_link"] = hit
        need = torch.is_grad_enabled() and (x.requires_grad or any(t.requires_grad for pair in loras for t in pair))
        return hpack, ops, scale, p, hit[2], need

    def _tail(self, o, x, gelu: str, next_tb=None):
        """x + to_out(o), then + ff(norm3(.)): one row-tile chain launch each way (HF.block_tail) on the bf16 path, else
        the launch-per-stage form (GEMM, LayerNorm, two GEMMs)."""
        act = "gelu_tanh" if self.ff.net[0].approximate == "tanh" else gelu
        if HF.can_block_tail(x, self.ff.net[0].proj.out_features, self.attn1.to_out[0].in_features) and o.dtype == x.dtype:
            pack = self._tail_pack()
            if pack is not None:
                return HF.block_tail(o, x, pack, act, self._link(pack, next_tb, x))
        x = hip_linear(self.attn1.to_out[0], o, residual=x)
        x, y = hip_layernorm_fork(self.norm3, x)
        return hip_ffn(self.ff.net[0].proj, self.ff.net[2], y, act, residual=x)


class ConditionalDecoder(nn.Module):
    """U-Net1D flow-matching estimator (cosyvoice/flow/decoder.py:88-291 == modules.py:886-1106,
    prompt isolation off).  Channel-last fast path: ``forward_cl``; ``forward`` keeps the
    reference's (B,C,T) signature."""

    def __init__(self, in_channels, out_channels, channels=(256, 256), dropout=0.05, attention_head_dim=64,
                 n_blocks=1, num_mid_blocks=2, num_heads=4, act_fn="gelu"):
        super().__init__()
        channels = tuple(channels)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.prompt_isolation_enabled = False
        self.prompt_isolation_len = 0
        self.time_embeddings = SinusoidalPosEmb(in_channels)
        ted = channels[0] * 4
        self.time_mlp = TimestepEmbedding(in_channels, ted, act_fn="silu")
        self.down_blocks, self.mid_blocks, self.up_blocks = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()

        def tbs(ch):
            return nn.ModuleList([BasicTransformerBlock(ch, num_heads, attention_head_dim, dropout, act_fn)
                                  for _ in range(n_blocks)])
        oc = in_channels
        for i, ch in enumerate(channels):
            ic, oc = oc, ch
            last = i == len(channels) - 1
            self.down_blocks.append(nn.ModuleList([ResnetBlock1D(ic, oc, ted), tbs(oc),
                                                   Downsample1D(oc) if not last else nn.Conv1d(oc, oc, 3, padding=1)]))
        for _ in range(num_mid_blocks):
            self.mid_blocks.append(nn.ModuleList([ResnetBlock1D(channels[-1], channels[-1], ted), tbs(channels[-1])]))
        ch2 = channels[::-1] + (channels[0],)
        for i in range(len(ch2) - 1):
            ic, oc = ch2[i] * 2, ch2[i + 1]
            last = i == len(ch2) - 2
            self.up_blocks.append(nn.ModuleList([ResnetBlock1D(ic, oc, ted), tbs(oc),
                                                 Upsample1D(oc) if not last else nn.Conv1d(oc, oc, 3, padding=1)]))
        self.final_block = Block1D(ch2[-1], ch2[-1])
        self.final_proj = nn.Conv1d(ch2[-1], out_channels, 1)
        self._initialize_weights()

    def _initialize_weights(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv1d, nn.Linear)):
                nn.init.kaiming_normal_(m.weight, nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.GroupNorm):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward_cl(self, xin, t, B: int, T: int, length, gelu: str = "gelu_erf", t_true=None):
        """xin [B*T, in_channels] (already [y|mu|spk|cond] packed), t [B] fp32, length int32 [B]
        -> [B*T, out_channels] (masked).
        t_true (device int32 [levels], optional): T is padded to a shape bucket and t_true[l] holds the exact batch's frame
        count at U-Net level l (T_max, ceil(T_max / 2), ...): the GroupNorms normalise over those frames only (cvft.h t_eff);
        everything else on the path is masked by `length` already."""
        dtype = xin.dtype
        plen = self.prompt_isolation_len if self.prompt_isolation_enabled else 0

        def iso(Tl: int) -> int:
            """modules.py:1033-1042: the prompt / target split rescaled to this U-Net level (block-diagonal attention)."""
            if plen <= 0:
                return 0
            sp = max(1, int(plen * (Tl / T)))
            return sp if sp < Tl else 0
        with torch.no_grad():
            temb = self.time_mlp(self.time_embeddings(t, dtype=dtype))
            temb_mish = HF.act_fwd(temb, "mish")
        x = xin
        hiddens: List[Tuple[torch.Tensor, int, torch.Tensor, int]] = []
        Tc, lc, lv = T, length, 0
        te = (lambda l: None) if t_true is None else (lambda l: t_true[l:l + 1])
        for resnet, tblocks, down in self.down_blocks:
            x = resnet(x, B, Tc, lc, temb_mish, te(lv))
            for i, tb in enumerate(tblocks):
                x = tb(x, B, Tc, lc, gelu, iso(Tc), tblocks[i + 1] if i + 1 < len(tblocks) else None)
            hiddens.append((x, Tc, lc, lv))
            if isinstance(down, Downsample1D):
                pk = conv_pack(down.conv, dtype)
                Tn = pk.out_len(Tc)
                x = HF.conv1d(x, pk, B, Tc, Tn, in_len=lc)
                Tc, lc, lv = Tn, (lc + 1) // 2, lv + 1          # mask[:, :, ::2]
            else:
                x = HF.conv1d(x, conv_pack(down, dtype), B, Tc, in_len=lc)
                # reference appends mask_down[:, :, ::2] then drops it (masks = masks[:-1])
        for resnet, tblocks in self.mid_blocks:
            x = resnet(x, B, Tc, lc, temb_mish, te(lv))
            for i, tb in enumerate(tblocks):
                x = tb(x, B, Tc, lc, gelu, iso(Tc), tblocks[i + 1] if i + 1 < len(tblocks) else None)
        for resnet, tblocks, up in self.up_blocks:
            skip, Ts, ls, lvs = hiddens.pop()
            assert Ts == Tc, (Ts, Tc)
            x = torch.cat([x, skip], dim=1)
            x = resnet(x, B, Ts, ls, temb_mish, te(lvs))
            for i, tb in enumerate(tblocks):
                x = tb(x, B, Ts, ls, gelu, iso(Ts), tblocks[i + 1] if i + 1 < len(tblocks) else None)
            if isinstance(up, Upsample1D):
                Tn = hiddens[-1][1]                 # cropped to the next skip's length
                x = HF.conv1d(x, conv_pack(up.conv, dtype), B, Ts, Tn, in_len=ls)
                Tc, lc, lv = Tn, hiddens[-1][2], hiddens[-1][3]
            else:
                x = HF.conv1d(x, conv_pack(up, dtype), B, Ts, in_len=ls)
                Tc, lc, lv = Ts, ls, lvs
        x = self.final_block(x, B, Tc, lc, t_eff=te(lv))
        return HF.conv1d(x, conv_pack(self.final_proj, dtype), B, Tc, in_len=lc, out_len=lc)

    def forward(self, x, mask, mu, t, spks=None, cond=None, dtype: Optional[torch.dtype] = None, gelu="gelu_erf"):
        """Reference signature: x,mu,cond (B,80,T); mask (B,1,T); t (B,); spks (B,80) -> (B,80,T)."""
        B, _, T = x.shape
        dtype = x.dtype if dtype is None else dtype
        parts = [x, mu]
        if spks is not None:
            parts.append(spks.unsqueeze(-1).expand(-1, -1, T))
        if cond is not None:
            parts.append(cond)
        xin = torch.cat(parts, dim=1).transpose(1, 2).reshape(B * T, -1).to(dtype).contiguous()
        length = mask.reshape(B, T).sum(dim=1).to(torch.int32)
        out = self.forward_cl(xin, t.reshape(-1).float(), B, T, length, gelu)
        return out.reshape(B, T, -1).transpose(1, 2).to(x.dtype)


# =================================================================================
# Encoders (cosyvoice/transformer/*; twin modules.py:382-793)
# =================================================================================
class EspnetRelPositionalEncoding(nn.Module):
    """embedding.py:201-302.  The (1, 2L-1, d) sin/cos table is a host-built constant (the
    reference builds ``self.pe`` on the CPU in __init__ as well); x*sqrt(d) is fused into the
    preceding LayerNorm kernel as its post-scale."""

    def __init__(self, d_model: int, dropout_rate: float = 0.0, max_len: int = 5000):
        super().__init__()
        self.d_model = d_model
        self.xscale = math.sqrt(d_model)
        self.dropout_rate = dropout_rate
        self._cache = {}

    def table(self, L: int, device, dtype) -> torch.Tensor:
        key = (L, str(device), dtype)
        if key not in self._cache:
            d = self.d_model
            pos = (L - 1 - torch.arange(0, 2 * L - 1, dtype=torch.float32)).unsqueeze(1)
            div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
            pe = torch.zeros(2 * L - 1, d, dtype=torch.float32)
            pe[:, 0::2] = torch.sin(pos * div)
            pe[:, 1::2] = torch.cos(pos * div)
            # (never evicted: a captured hipGraph holds the address of the table it was captured with; ~0.5 MB per length)
            self._cache[key] = pe.to(device=device, dtype=dtype).contiguous()
            self._cache[key]._cvft_const = True      # (a frozen linear_pos caches its projection of this table)
        return self._cache[key]


class LinearNoSubsampling(nn.Module):
    """subsampling.py:69-113 (legacy=False) / 338-383 (legacy=True: + ReLU)."""

    def __init__(self, idim, odim, dropout_rate, pos_enc, legacy=False):
        super().__init__()
        mods = [nn.Linear(idim, odim), nn.LayerNorm(odim, eps=1e-5), nn.Dropout(dropout_rate)]
        if legacy:
            mods.append(nn.ReLU())
        self.out = nn.Sequential(*mods)
        self.pos_enc = pos_enc
        self.legacy = legacy


class RelPositionMultiHeadedAttention(nn.Module):
    """attention.py:200-330 parameters; compute = fused rel-pos flash kernel (incl. the attention-probability dropout)."""

    def __init__(self, n_head, n_feat, dropout_rate, key_bias=True):
        super().__init__()
        assert n_feat % n_head == 0
        self.d_k, self.h = n_feat // n_head, n_head
        if self.d_k != 64:
            raise NotImplementedError("fused attention kernels are specialised to head_dim 64")
        self.linear_q = nn.Linear(n_feat, n_feat)
        self.linear_k = nn.Linear(n_feat, n_feat, bias=key_bias)
        self.linear_v = nn.Linear(n_feat, n_feat)
        self.linear_out = nn.Linear(n_feat, n_feat)
        self.dropout_rate = dropout_rate
        self.linear_pos = nn.Linear(n_feat, n_feat, bias=False)
        self.pos_bias_u = nn.Parameter(torch.Tensor(self.h, self.d_k))
        self.pos_bias_v = nn.Parameter(torch.Tensor(self.h, self.d_k))
        nn.init.xavier_uniform_(self.pos_bias_u)
        nn.init.xavier_uniform_(self.pos_bias_v)

    def forward(self, y, residual, pos_emb, B, L, length, causal, out_dropout: float = 0.0):
        q, k, v = hip_qkv(self.linear_q, self.linear_k, self.linear_v, y)
        p = None
        w = self.linear_pos.weight if isinstance(self.linear_pos, nn.Linear) else None
        if w is not None and not w.requires_grad and getattr(pos_emb, "_cvft_const", False):
            # frozen projection of a constant table (eval mode: no positional dropout): once per table, not once per layer
            # and step.  Entries are never evicted -- a captured hipGraph may hold the address of one -- so the cache
            # simply stops growing at 16 tables.
            cache = self.__dict__.setdefault("_cvft_pcache", {})
            key = (pos_emb.data_ptr(), tuple(pos_emb.shape), pos_emb.dtype, w._version, w.data_ptr())
            p = cache.get(key)
            if p is None and len(cache) < 16:
                p = cache[key] = hip_linear(self.linear_pos, pos_emb).detach()
        if p is None:
            p = hip_linear(self.linear_pos, pos_emb)
        o = HF.attn_relpos(q, k, v, p, _f32(self.pos_bias_u), _f32(self.pos_bias_v), B, self.h, L, length, causal,
                           1.0 / math.sqrt(self.d_k), dropout_p=self.dropout_rate if self.training else 0.0)
        # x = residual + dropout(linear_out(.))  (encoder_layer.py:95 / 205)
        return hip_linear(self.linear_out, o, residual=residual, out_drop=out_dropout)


class PositionwiseFeedForward(nn.Module):
    def __init__(self, idim, hidden_units, dropout_rate, activation: str):
        super().__init__()
        self.w_1 = nn.Linear(idim, hidden_units)
        self.activation = activation          # "relu" | "silu"
        self.dropout_rate = dropout_rate
        self.w_2 = nn.Linear(hidden_units, idim)

    def forward(self, y, residual, out_dropout: float = 0.0):
        # w_2(dropout(act(w_1 x)))  (positionwise_feed_forward.py:54), then residual + dropout(.) (encoder_layer.py:104 / 234)
        return hip_ffn(self.w_1, self.w_2, y, self.activation, residual=residual,
                       inner_drop=self.dropout_rate if self.training else 0.0, out_drop=out_dropout if self.training else 0.0)


class EncoderLayer(nn.Module):
    """TransformerEncoderLayer (encoder_layer.py:27-106: norm1/norm2) or ConformerEncoderLayer without
    macaron / cnn module (encoder_layer.py:109-236: norm_mha/norm_ff), pre-norm."""

    def __init__(self, size, self_attn, feed_forward, conformer: bool, eps: float, dropout_rate: float = 0.0):
        super().__init__()
        self.self_attn = self_attn
        self.feed_forward = feed_forward
        self.conformer = conformer
        self.dropout_rate = dropout_rate
        if conformer:
            self.norm_ff = nn.LayerNorm(size, eps=eps)
            self.norm_mha = nn.LayerNorm(size, eps=eps)
        else:
            self.norm1 = nn.LayerNorm(size, eps=eps)
            self.norm2 = nn.LayerNorm(size, eps=eps)

    def forward(self, x, pos_emb, B, L, length, causal, eps):
        n_att, n_ff = (self.norm_mha, self.norm_ff) if self.conformer else (self.norm1, self.norm2)
        pd = self.dropout_rate if self.training else 0.0
        at, ff = self.self_attn, self.feed_forward
        x, y = hip_layernorm_fork(n_att, x, eps, consumers=(at.linear_q, at.linear_k, at.linear_v))
        x = self.self_attn(y, x, pos_emb, B, L, length, causal, out_dropout=pd)
        x, y = hip_layernorm_fork(n_ff, x, eps, consumers=(ff.w_1,) if self.training else None)
        return self.feed_forward(y, x, out_dropout=pd)


class RelPosEncoder(nn.Module):
    """BaseEncoder.forward (encoder.py:111-170) for the two encoder flavours CosyVoice-300M uses:
    ConformerEncoder(no cnn, no macaron, swish) and TransformerEncoder(relu), both with
    rel_pos_espnet + rel_selfattn and optional static_chunk_size=1 (causal)."""

    def __init__(self, input_size, output_size=256, attention_heads=4, linear_units=2048, num_blocks=6,
                 dropout_rate=0.1, positional_dropout_rate=0.1, attention_dropout_rate=0.0, input_layer="linear",
                 kind="conformer", static_chunk_size=0, key_bias=True, ln_eps=1e-12):
        super().__init__()
        assert kind in ("conformer", "transformer") and input_layer in ("linear", "linear_legacy")
        self._output_size = output_size
        self.kind, self.static_chunk_size = kind, static_chunk_size
        pos = EspnetRelPositionalEncoding(output_size, positional_dropout_rate)
        self.embed = LinearNoSubsampling(input_size, output_size, dropout_rate, pos, legacy=(input_layer == "linear_legacy"))
        self.normalize_before = True
        self.after_norm = nn.LayerNorm(output_size, eps=1e-5)
        act = "silu" if kind == "conformer" else "relu"
        self.encoders = nn.ModuleList([
            EncoderLayer(output_size,
                         RelPositionMultiHeadedAttention(attention_heads, output_size, attention_dropout_rate, key_bias),
                         PositionwiseFeedForward(output_size, linear_units, dropout_rate, act),
                         conformer=(kind == "conformer"), eps=ln_eps, dropout_rate=dropout_rate)
            for _ in range(num_blocks)])

    def output_size(self) -> int:
        return self._output_size

    def forward_cl(self, xs, B: int, L: int, length, num: Numerics, causal: Optional[bool] = None):
        """xs [B*L, input_size] -> [B*L, d]."""
        causal = (self.static_chunk_size > 0) if causal is None else causal
        d = self._output_size
        x = hip_linear(self.embed.out[0], xs, dtype=num.dtype)
        x = hip_layernorm(self.embed.out[1], x, relu=self.embed.legacy, post_scale=math.sqrt(d) if num.xscale else 1.0)
        pos_emb = self.embed.pos_enc.table(L, x.device, num.dtype)
        if self.training:
            # Dropout after the input LayerNorm (subsampling.py:84) and the two of the positional encoding
            # (embedding.py:285-288: on x*sqrt(d) and on pos_emb); ReLU / the sqrt(d) scale commute with the masks
            x = HF.dropout_add(x, self.embed.out[2].p)
            x = HF.dropout_add(x, self.embed.pos_enc.dropout_rate)
            pos_emb = HF.dropout_add(pos_emb, self.embed.pos_enc.dropout_rate)
        for layer in self.encoders:
            x = layer(x, pos_emb, B, L, length, causal, num.enc_ln_eps)
        return hip_layernorm(self.after_norm, x)

    def forward(self, xs, xs_lens, decoding_chunk_size: int = 0, num_decoding_left_chunks: int = -1,
                num: Optional[Numerics] = None):
        """Reference signature: xs (B,L,D), xs_lens (B,) -> (B,L,d), masks (B,1,L) bool."""
        num = num or Numerics(dtype=xs.dtype)
        B, L, D = xs.shape
        length = to_len(xs_lens, xs.device)
        out = self.forward_cl(xs.reshape(B * L, D), B, L, length, num)
        masks = (torch.arange(L, device=xs.device).unsqueeze(0) < length.unsqueeze(1)).unsqueeze(1)
        return out.reshape(B, L, -1), masks


class InterpolateRegulator(nn.Module):
    """length_regulator.py:21-50 == modules.py:800-821."""

    def __init__(self, channels: int, sampling_ratios: Tuple, out_channels: int = None, groups: int = 1):
        super().__init__()
        self.sampling_ratios = sampling_ratios
        out_channels = out_channels or channels
        model = nn.ModuleList([])
        for _ in sampling_ratios:
            model.extend([nn.Conv1d(channels, channels, 3, 1, 1), nn.GroupNorm(groups, channels), nn.Mish()])
        model.append(nn.Conv1d(channels, out_channels, 1, 1))
        self.model = nn.Sequential(*model)

    def inference_cl(self, x1, x2, mel_len1: int, mel_len2: int, input_frame_rate: int = 50):
        """length_regulator.py:52-70 == modules.py:823-838 (batch 1): x1 [L1, C] prompt part, x2 [L2, C] target part ->
        [mel_len1 + mel_len2, C].  The parts are interpolated separately, the target's first / last 20 tokens at the
        nominal rate and its middle taking up the slack, so the prompt/target seam falls on an exact frame."""
        up = lambda t, n: HF.interp_linear(t.contiguous(), 1, t.shape[0], n)
        if x2.shape[0] > 40:
            e = int(20 / input_frame_rate * 22050 / 256)
            x2 = torch.cat([up(x2[:20], e), up(x2[20:-20], mel_len2 - 2 * e), up(x2[-20:], e)], dim=0)
        else:
            x2 = up(x2, mel_len2)
        x = torch.cat([up(x1, mel_len1), x2], dim=0) if x1.shape[0] != 0 else x2
        return self._stack_cl(x.contiguous(), 1, mel_len1 + mel_len2, None)

    def inference(self, x1, x2, mel_len1, mel_len2, input_frame_rate=50):
        out = self.inference_cl(x1[0], x2[0], int(mel_len1), int(mel_len2), input_frame_rate)
        return out.unsqueeze(0), mel_len1 + mel_len2

    def forward_cl(self, x, B: int, Lin: int, T: int, ylen, eff=None):
        """x [B*Lin, C] -> [B*T, C] masked by ylen (int32 [B]).
        eff (device int32 [2], optional): Lin / T are padded to shape buckets and eff = the exact batch's (Lt_max, T_max) --
        the interpolation takes its scale and clamp from eff, the GroupNorms normalise over T_max frames, frames beyond
        are zeros throughout (what the convolutions of the exact-shape batch see as their zero padding)."""
        return self._stack_cl(HF.interp_linear(x, B, Lin, T, eff), B, T, ylen, None if eff is None else eff[1:2])

    def _stack_cl(self, x, B: int, T: int, ylen, t_eff=None):
        mods = list(self.model)
        i = 0
        while i + 2 < len(mods):
            conv, gn = mods[i], mods[i + 1]
            x = HF.conv1d(x, conv_pack(conv, x.dtype), B, T)
            x = HF.groupnorm_mish(x, _f32(gn.weight), _f32(gn.bias), B, T, gn.num_groups, gn.eps, None, None, True, t_eff)
            i += 3
        return HF.conv1d(x, conv_pack(mods[-1], x.dtype), B, T, out_len=ylen)

    def forward(self, x, ylens=None):
        B, Lin, Cc = x.shape
        T = int(ylens.max())
        out = self.forward_cl(x.reshape(B * Lin, Cc), B, Lin, T, to_len(ylens, x.device))
        return out.reshape(B, T, -1), ylens


class ConvolutionModule(nn.Module):
    """Conformer convolution module (convolution.py:24-145, layer_norm variant): pointwise 1x1 -> GLU ->
    depthwise Conv1d(k, groups=C) -> LayerNorm -> act -> pointwise 1x1.  Not executed by the
    CosyVoice-300M config (use_cnn_module=False); provided op-complete with HIP kernels."""

    def __init__(self, channels, kernel_size=15, activation="silu", norm="layer_norm", causal=False):
        super().__init__()
        if norm != "layer_norm":
            raise NotImplementedError("batch_norm variant is not built")
        self.pointwise_conv1 = nn.Conv1d(channels, 2 * channels, 1)
        self.lorder = kernel_size - 1 if causal else 0
        pad = 0 if causal else (kernel_size - 1) // 2
        self.depthwise_conv = nn.Conv1d(channels, channels, kernel_size, padding=pad, groups=channels)
        self.norm = nn.LayerNorm(channels)
        self.pointwise_conv2 = nn.Conv1d(channels, channels, 1)
        self.activation = activation

    def forward_cl(self, x, B: int, T: int, length):
        Cc = x.shape[1]
        h = HF.conv1d(x, conv_pack(self.pointwise_conv1, x.dtype), B, T, in_len=length)
        a, g = h[:, :Cc], h[:, Cc:]
        h = a * torch.sigmoid(g.float()).to(a.dtype)                       # GLU (glue; op not on the 300M path)
        dw = self.depthwise_conv
        pad_left = self.lorder if self.lorder > 0 else (dw.kernel_size[0] - 1) // 2
        h = HF.dwconv1d(h, _f32(dw.weight).reshape(Cc, -1).contiguous(), _f32(dw.bias), B, T, pad_left)
        h = hip_layernorm(self.norm, h)
        h = HF.ActFn.apply(h, self.activation)
        return HF.conv1d(h, conv_pack(self.pointwise_conv2, x.dtype), B, T, out_len=length)
