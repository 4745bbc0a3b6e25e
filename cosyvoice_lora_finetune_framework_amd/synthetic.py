"""Synthetic (text, speech-token, mel, x-vector) batches in the reference's batch format.

The batch dict has the keys/dtypes/padding the reference's ``collate_fn`` produces
(reference dataset.py:549-596): ``speech_token (B,Lt) int64 pad 0``, ``speech_token_len``,
``speech_feat (B,T,80) fp32 raw log-mel pad -11.5``, ``speech_feat_len``,
``embedding (B,192)``, ``text_token (B,Lx) int64 pad 0``, ``text_token_len``.
Shapes follow SURVEY.md section 8(d): Lt = floor(T*50*256/22050).
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch


def speech_tokens_for_frames(T: int) -> int:
    """50 Hz speech tokens vs 22050/256 Hz mel frames (reference flow_model.py:530-536)."""
    return int(T * 50 * 256 / 22050)


def synth_batch(feat_lens: Sequence[int], text_lens: Optional[Sequence[int]] = None,
                token_lens: Optional[Sequence[int]] = None, seed: int = 1234,
                text_vocab: int = 51866, speech_vocab: int = 4096, n_mels: int = 80,
                spk_dim: int = 192) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    B = len(feat_lens)
    if token_lens is None:
        token_lens = [speech_tokens_for_frames(t) for t in feat_lens]
    if text_lens is None:
        text_lens = [40] * B
    T, Lt, Lx = max(feat_lens), max(token_lens), max(text_lens)
    feat = torch.randn(B, T, n_mels, generator=g) * 2.0 - 6.0
    tok = torch.randint(0, speech_vocab, (B, Lt), generator=g, dtype=torch.int64)
    txt = torch.randint(0, text_vocab, (B, Lx), generator=g, dtype=torch.int64)
    emb = torch.randn(B, spk_dim, generator=g)
    for i in range(B):
        feat[i, feat_lens[i]:] = -11.5
        tok[i, token_lens[i]:] = 0
        txt[i, text_lens[i]:] = 0
    return {
        "speech_token": tok,
        "speech_token_len": torch.tensor(list(token_lens), dtype=torch.int64),
        "speech_feat": feat,
        "speech_feat_len": torch.tensor(list(feat_lens), dtype=torch.int64),
        "embedding": emb,
        "text_token": txt,
        "text_token_len": torch.tensor(list(text_lens), dtype=torch.int64),
    }


def cfm_draws(B: int, T: int, seed: int, n_mels: int = 80) -> Dict[str, torch.Tensor]:
    """The three CFM random draws in the reference's order (reference
    cosyvoice/flow/flow_matching.py:175-186): rand(B,1,1), randn(B,80,T), rand(B).
    With ``torch.manual_seed(seed)`` the reference's CPU path draws exactly these."""
    g = torch.Generator().manual_seed(seed)
    t_raw = torch.rand([B, 1, 1], generator=g)
    z = torch.randn(B, n_mels, T, generator=g)
    cfg_rand = torch.rand(B, generator=g)
    return {"t_raw": t_raw, "z": z, "cfg_rand": cfg_rand}
