"""LoRA adapters -- drop-in for the reference's ``lora.py`` (same names, argument order,
defaults, state-dict keys and return values; reference lora.py:18-323), with the forward
pass running as ONE fused MI355X GEMM (``y = x W^T + b + (alpha/r) (x A^T) B^T``: MFMA main
loop + rank-r side path appended to the K loop) instead of 3 ``F.linear`` + scale + add.

Differences from the reference, all deliberate:
  * compute happens in libcvft (HIP); CPU tensors raise -- there is no eager fallback;
  * ``lora_dropout`` is accepted and stored, and applied (to the side-path input only, as in
    lora.py:70) when the module is in training mode and p > 0: the mask is counter-based and
    drawn INSIDE the rank-side kernel (a pure function of the device-side step seed, the call
    site and the element index; ``hipops.functional.dropout_begin_step``), re-derived in
    backward instead of stored.  fp32 / odd shapes fall back to ``nn.Dropout``'s torch mask.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn


class LoRALinear(nn.Module):
    """reference lora.py:18-76.  ``lora_A (r,in)`` kaiming-uniform(a=sqrt 5), ``lora_B (out,r)``
    normal(0, 0.01) (NOT zeros), ``scaling = lora_alpha / r``; original weight/bias frozen."""

    def __init__(self, original_layer: nn.Linear, r: int = 8, lora_alpha: int = 16, lora_dropout: float = 0.1):
        super().__init__()
        self.original_layer = original_layer
        self.r = r
        self.lora_alpha = lora_alpha
        self.scaling = lora_alpha / r
        in_features, out_features = original_layer.in_features, original_layer.out_features
        self.original_layer.weight.requires_grad = False
        if self.original_layer.bias is not None:
            self.original_layer.bias.requires_grad = False
        dev = original_layer.weight.device
        self.lora_A = nn.Parameter(torch.zeros(r, in_features, device=dev))
        self.lora_B = nn.Parameter(torch.zeros(out_features, r, device=dev))
        self.lora_dropout = nn.Dropout(p=lora_dropout) if lora_dropout > 0 else nn.Identity()
        nn.init.kaiming_uniform_(self.lora_A, a=math.sqrt(5))
        nn.init.normal_(self.lora_B, mean=0.0, std=0.01)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        from .modules import hip_linear
        shp = x.shape
        y = hip_linear(self, x.reshape(-1, shp[-1]))
        return y.reshape(*shp[:-1], y.shape[-1])


class LoRAConv1d(nn.Module):
    """reference lora.py:79-131 (1x1 Conv1d == Linear over channels; never instantiated by the
    shipped target lists).  Input (B, C, T) like nn.Conv1d."""

    def __init__(self, original_layer: nn.Conv1d, r: int = 8, lora_alpha: int = 16, lora_dropout: float = 0.1):
        super().__init__()
        self.original_layer = original_layer
        self.r = r
        self.lora_alpha = lora_alpha
        self.scaling = lora_alpha / r
        in_channels, out_channels = original_layer.in_channels, original_layer.out_channels
        self.original_layer.weight.requires_grad = False
        if self.original_layer.bias is not None:
            self.original_layer.bias.requires_grad = False
        self.lora_A = nn.Conv1d(in_channels, r, kernel_size=1, bias=False)
        self.lora_B = nn.Conv1d(r, out_channels, kernel_size=1, bias=False)
        self.lora_dropout = nn.Dropout(p=lora_dropout) if lora_dropout > 0 else nn.Identity()
        nn.init.kaiming_uniform_(self.lora_A.weight, a=math.sqrt(5))
        nn.init.normal_(self.lora_B.weight, mean=0.0, std=0.01)
        self.to(original_layer.weight.device)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        from .modules import hip_linear
        B, Cc, T = x.shape
        y = hip_linear(self, x.transpose(1, 2).reshape(B * T, Cc))
        return y.reshape(B, T, -1).transpose(1, 2)


DEFAULT_TARGET_MODULES = ['to_q', 'to_k', 'to_v', 'linear_q', 'linear_k', 'linear_v', 'linear_out', 'w_1', 'w_2',
                          'linear_pos']   # reference lora.py:155-166


def apply_lora_to_model(model: nn.Module, r: int = 8, lora_alpha: int = 16, lora_dropout: float = 0.1,
                        target_modules: Optional[List[str]] = None) -> Dict[str, int]:
    """reference lora.py:134-227: wrap every nn.Linear / 1x1 nn.Conv1d child whose *own attribute
    name* contains a target string; then freeze every parameter without 'lora_' in its name."""
    if target_modules is None:
        target_modules = list(DEFAULT_TARGET_MODULES)
    targets = set(target_modules)
    replaced = 0
    total_lora = 0
    original_params = sum(p.numel() for p in model.parameters())

    def visit(parent: nn.Module):
        nonlocal replaced, total_lora
        for name, child in list(parent.named_children()):
            if any(t in name for t in targets):
                if isinstance(child, nn.Linear):
                    wrapped = LoRALinear(child, r=r, lora_alpha=lora_alpha, lora_dropout=lora_dropout)
                    setattr(parent, name, wrapped)
                    replaced += 1
                    total_lora += wrapped.lora_A.numel() + wrapped.lora_B.numel()
                elif isinstance(child, nn.Conv1d) and child.kernel_size[0] == 1:
                    wrapped = LoRAConv1d(child, r=r, lora_alpha=lora_alpha, lora_dropout=lora_dropout)
                    setattr(parent, name, wrapped)
                    replaced += 1
                    total_lora += wrapped.lora_A.weight.numel() + wrapped.lora_B.weight.numel()
            visit(child)

    visit(model)
    for name, param in model.named_parameters():
        if 'lora_' not in name:
            param.requires_grad = False
    trainable = sum(p.numel() for p in model.parameters() if p.requires_grad)
    return {'replaced_layers': replaced, 'original_params': original_params, 'lora_params': total_lora,
            'trainable_params': trainable, 'trainable_ratio': trainable / original_params * 100}


def get_lora_state_dict(model: nn.Module) -> Dict[str, torch.Tensor]:
    """reference lora.py:230-236."""
    return {n: p.data.clone() for n, p in model.named_parameters() if 'lora_' in n}


def save_lora_weights(model: nn.Module, path: str):
    """reference lora.py:239-243."""
    sd = get_lora_state_dict(model)
    torch.save(sd, path)
    print(f"Saved LoRA weights: {len(sd)} tensors to {path}")


def load_lora_weights(model: nn.Module, path: str):
    """reference lora.py:246-256."""
    sd = torch.load(path, map_location='cpu', weights_only=True)
    msd = model.state_dict()
    for name, param in sd.items():
        if name in msd:
            msd[name].copy_(param)
    print(f"Loaded LoRA weights: {len(sd)} tensors from {path}")


def merge_lora_weights(model: nn.Module):
    """reference lora.py:259-281: W += (B @ A) * scaling, in place on the frozen weight."""
    for _, module in model.named_modules():
        if isinstance(module, LoRALinear):
            with torch.no_grad():
                module.original_layer.weight.add_(module.lora_B @ module.lora_A * module.scaling)
        elif isinstance(module, LoRAConv1d):
            with torch.no_grad():
                delta = torch.einsum('ori,ric->oic', module.lora_B.weight, module.lora_A.weight) * module.scaling
                module.original_layer.weight.add_(delta)
    print("LoRA weights merged into original model")


def get_merged_state_dict(model: nn.Module) -> dict:
    """reference lora.py:284-323: merge, then export in the ORIGINAL key format
    (``<path>.weight/.bias`` for wrapped layers; no lora_A/lora_B/original_layer keys)."""
    merge_lora_weights(model)
    out = {}
    for name, module in model.named_modules():
        if isinstance(module, (LoRALinear, LoRAConv1d)):
            out[f"{name}.weight"] = module.original_layer.weight.data.clone()
            if module.original_layer.bias is not None:
                out[f"{name}.bias"] = module.original_layer.bias.data.clone()
    for name, param in model.named_parameters():
        if 'lora_A' in name or 'lora_B' in name or 'original_layer' in name:
            continue
        out[name] = param.data.clone()
    for name, buf in model.named_buffers():
        if 'lora_' not in name and 'original_layer' not in name:
            out[name] = buf.clone()
    print(f"Exported merged state_dict with {len(out)} keys")
    return out
