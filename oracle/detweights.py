"""Deterministic, construction-order-independent weight fill (TEST INFRASTRUCTURE).

The reference's pretrained ``llm.pt`` / ``flow.pt`` are not available offline and a
full-size state_dict (1.6 GB) cannot travel as a fixture.  Instead every tensor is a
pure function of (its key name, its shape, a seed): ``tools/make_golden.py`` loads these
weights into the *reference* modules to produce golden losses, and the GPU tests
regenerate the very same tensors on the GPU box and load them into the product model.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import torch


def det_tensor(name: str, shape: Tuple[int, ...], seed: int = 0) -> torch.Tensor:
    g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
    shape = tuple(int(s) for s in shape)
    if len(shape) == 0:
        return torch.zeros((), dtype=torch.float32)
    x = torch.randn(shape, generator=g, dtype=torch.float32)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "lora_B" or name.endswith("lora_B.weight"):
        return x * 0.05
    if len(shape) == 1:
        return 1.0 + 0.1 * x if leaf == "weight" else 0.05 * x
    if leaf in ("pos_bias_u", "pos_bias_v"):
        return 0.1 * x
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return x / (fan_in ** 0.5)


def det_state_dict(spec: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = 0) -> Dict[str, torch.Tensor]:
    """spec: iterable of (key, shape) -- e.g. ``[(k, v.shape) for k, v in model.state_dict().items()]``."""
    return {k: det_tensor(k, tuple(s), seed) for k, s in spec}
