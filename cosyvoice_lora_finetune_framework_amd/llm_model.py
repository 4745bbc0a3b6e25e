"""Text-token LLM (text -> speech-token LM) on the HIP path.

``TransformerLM`` keeps the attribute names / state-dict keys of the vendored
cosyvoice/llm/llm.py:32-95 that ``JointLLMFlowModel._forward_llm`` consumes
(llm_flow_model.py:118-179): text_embedding, text_encoder, text_encoder_affine_layer,
llm_embedding, llm, llm_decoder, speech_embedding, spk_embed_affine_layer, sos_eos, task_id,
speech_token_size.  Only the training forward (no-prompt) is built; AR sampling / Qwen2LM are
out of scope.

MI355X-first: the ragged ``[sos, spk, text, task, speech]`` concat + ``-1`` padding
(llm.py:88-95) is ONE row-gather kernel driven by an index map computed on the host from the
CPU-side length vectors (no device sync); the LM target never leaves int32 index form.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch
import torch.nn as nn

from .config import LLM_MODEL_CONFIG
from .hipops import functional as HF
from .modules import Numerics, RelPosEncoder, _cached, hip_linear, to_len
from .utils import IGNORE_ID


class TransformerLM(nn.Module):
    def __init__(self, text_encoder_input_size: int, llm_input_size: int, llm_output_size: int, text_token_size: int,
                 speech_token_size: int, text_encoder: nn.Module, llm: nn.Module, sampling=None,
                 length_normalized_loss: bool = True, lsm_weight: float = 0.0, spk_embed_dim: int = 192):
        super().__init__()
        if not length_normalized_loss:      # (denominator = batch size: the sub-batch / DP recombination is built on token counts)
            raise NotImplementedError("only the length-normalised CE (the CosyVoice-300M setting) is built")
        self.lsm_weight = float(lsm_weight)
        self.llm_input_size = llm_input_size
        self.speech_token_size = speech_token_size
        self.text_embedding = nn.Embedding(text_token_size, text_encoder_input_size)
        self.text_encoder = text_encoder
        self.text_encoder_affine_layer = nn.Linear(self.text_encoder.output_size(), llm_input_size)
        self.sos_eos = 0
        self.task_id = 1
        self.llm_embedding = nn.Embedding(2, llm_input_size)
        self.llm = llm
        self.llm_decoder = nn.Linear(llm_output_size, speech_token_size + 1)
        self.speech_embedding = nn.Embedding(speech_token_size, llm_input_size)
        self.spk_embed_affine_layer = nn.Linear(spk_embed_dim, llm_input_size)
        self.sampling = sampling
        self.numerics = Numerics()

    def _table(self, emb: nn.Embedding, dtype):
        w = emb.weight
        return w.detach() if w.dtype == dtype else _cached(emb, "tab", w, dtype, lambda: w.detach().to(dtype))

    @staticmethod
    def build_index_maps(text_len, speech_len, speech_token, B: int, Lx: int, Lt: int, eos: int, pad_to: int = 1, L_min: int = 0):
        """Host-side (CPU tensors only) ragged layout of llm.py:88-95 + llm_flow_model.py:129-139.
        Source rows: [sos, task | spk(B) | text(B*Lx) | speech(B*Lt)].  Returns (idx [B*L] int32,
        target [B*L] int32, lm_len [B] int32, L).  pad_to > 1 rounds L up (rows past an utterance's length are the
        reference's -1 padding either way: masked keys, ignored targets), so that steps of nearby lengths share one
        captured hipGraph."""
        tl = [int(v) for v in text_len.tolist()]
        sl = [int(v) for v in speech_len.tolist()]
        lens = [3 + a + b for a, b in zip(tl, sl)]
        L = max(-(-max(lens) // pad_to) * pad_to, L_min)              # (L_min: the LM length of the captured step this batch joins)
        idx = torch.full((B, L), -1, dtype=torch.int32)
        tgt = torch.full((B, L), IGNORE_ID, dtype=torch.int32)
        st = speech_token.to(torch.int32)
        for i in range(B):
            a, b = tl[i], sl[i]
            idx[i, 0] = 0
            idx[i, 1] = 2 + i
            idx[i, 2:2 + a] = torch.arange(2 + B + i * Lx, 2 + B + i * Lx + a, dtype=torch.int32)
            idx[i, 2 + a] = 1
            idx[i, 3 + a:3 + a + b] = torch.arange(2 + B + B * Lx + i * Lt, 2 + B + B * Lx + i * Lt + b, dtype=torch.int32)
            # target: IGNORE * (2 + text) , speech tokens, EOS   (position 2+a .. 2+a+b)
            tgt[i, 2 + a:2 + a + b] = st[i, :b]
            tgt[i, 2 + a + b] = eos
        return idx.reshape(-1), tgt.reshape(-1), torch.tensor(lens, dtype=torch.int32), L

    def forward_no_prompt(self, batch: dict, device) -> Dict[str, Any]:
        """llm_flow_model.py:109-179."""
        num = self.numerics
        dt = num.dtype
        text = batch['text_token']
        B, Lx = text.shape
        sp = batch['speech_token']
        Lt = sp.shape[1]
        if '_lm_maps' in batch:                       # prepared once on the host (prepare_batch): no sync, graph-safe
            idx, tgt, lm_len, L = batch['_lm_maps']
        else:
            idx, tgt, lm_len, L = self.build_index_maps(batch['text_token_len'].cpu(), batch['speech_token_len'].cpu(),
                                                        sp.cpu(), B, Lx, Lt, self.speech_token_size)
            idx, tgt, lm_len = idx.to(device), tgt.to(device), lm_len.to(device)
        text_len = to_len(batch['text_token_len'], device)
        with torch.no_grad():
            temb = HF.embed_gather(text.to(device), self._table(self.text_embedding, dt))
            semb = HF.embed_gather(sp.to(device), self._table(self.speech_embedding, dt))
            spk = hip_linear(self.spk_embed_affine_layer, HF.l2norm_rows(batch['embedding'].to(device), dt))
            special = self._table(self.llm_embedding, dt)
        enc = self.text_encoder.forward_cl(temb, B, Lx, text_len, num, causal=True)   # encode(): decoding_chunk_size=1
        enc = hip_linear(self.text_encoder_affine_layer, enc)
        src = torch.cat([special, spk, enc, semb], dim=0)
        lm_in = HF.gather_rows(src, idx, float(IGNORE_ID))
        out = self.llm.forward_cl(lm_in, B, L, lm_len, num, causal=True)
        logits = hip_linear(self.llm_decoder, out)
        loss, acc = HF.cross_entropy(logits, tgt, self.lsm_weight)
        return {'loss': loss, 'acc': acc}

    def forward(self, batch: dict, device) -> Dict[str, Any]:
        return self.forward_no_prompt(batch, device)

    def prepare_batch(self, batch: dict, device, lm_pad: int = 1, lm_min: int = 0) -> dict:
        """Host-side preparation of one batch: index maps + H2D copies, done once per batch outside the
        (graph-capturable) step."""
        B, Lx = batch['text_token'].shape
        Lt = batch['speech_token'].shape[1]
        idx, tgt, lm_len, L = self.build_index_maps(batch['text_token_len'].cpu(), batch['speech_token_len'].cpu(),
                                                    batch['speech_token'].cpu(), B, Lx, Lt, self.speech_token_size, lm_pad, lm_min)
        return {'_lm_maps': (idx.to(device), tgt.to(device), lm_len.to(device), L)}


def build_llm_model(pretrained_path: Optional[str] = None, device: str = 'cuda', numerics: Optional[Numerics] = None,
                    text_encoder_input_size: int = 512, llm_input_size: int = 1024, llm_output_size: int = 1024,
                    text_token_size: int = 51866, speech_token_size: int = 4096, spk_embed_dim: int = 192,
                    attention_heads: int = 16, linear_units: int = 4096, text_encoder_blocks: int = 6,
                    llm_blocks: int = 14) -> TransformerLM:
    """CosyVoice-300M TransformerLM (public upstream cosyvoice.yaml dims; config.LLM_MODEL_CONFIG)."""
    numerics = numerics or Numerics()
    te = RelPosEncoder(input_size=text_encoder_input_size, output_size=llm_input_size, attention_heads=attention_heads,
                       linear_units=linear_units, num_blocks=text_encoder_blocks, dropout_rate=0.1,
                       positional_dropout_rate=0.1, attention_dropout_rate=0.0, input_layer="linear", kind="conformer",
                       static_chunk_size=1, ln_eps=numerics.enc_ln_eps)
    lm = RelPosEncoder(input_size=llm_input_size, output_size=llm_output_size, attention_heads=attention_heads,
                       linear_units=linear_units, num_blocks=llm_blocks, dropout_rate=0.1, positional_dropout_rate=0.1,
                       attention_dropout_rate=0.0, input_layer="linear_legacy", kind="transformer", static_chunk_size=1,
                       ln_eps=numerics.enc_ln_eps)
    model = TransformerLM(text_encoder_input_size, llm_input_size, llm_output_size, text_token_size, speech_token_size,
                          te, lm, sampling=None, length_normalized_loss=True, lsm_weight=0.0, spk_embed_dim=spk_embed_dim)
    model.numerics = numerics
    if pretrained_path is not None:
        import os
        wf = os.path.join(pretrained_path, 'llm.pt') if os.path.isdir(pretrained_path) else pretrained_path
        if os.path.exists(wf):
            model.load_state_dict(torch.load(wf, map_location='cpu', weights_only=True), strict=True)
        else:
            print(f"Warning: Weight file not found: {wf}\nUsing random initialization")
    return model.to(device)
