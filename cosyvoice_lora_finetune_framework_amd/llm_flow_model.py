"""Joint LLM + Flow LoRA model -- drop-in for the reference's llm_flow_model.py
(``JointLLMFlowModel`` :33-229, ``build_joint_model`` :232-310,
``get_joint_merged_state_dict`` :313-336): same constructor arguments, ``forward(batch, device)``
contract (dict of 0-d tensors ``loss / llm_loss / flow_loss / llm_acc``) and loss weighting.
The two sub-model forwards run on the HIP path (llm_model.py / flow_model.py)."""
from __future__ import annotations

import contextlib
import os

from typing import Any, Dict, Optional

import torch
import torch.nn as nn

from .config import JOINT_TRAINING_CONFIG, MEL_MEAN, MEL_STD, PRETRAINED_MODEL_DIR
from .flow_model import build_flow_model
from .llm_model import build_llm_model
from .lora import apply_lora_to_model
from .hipops import functional as HF
from .modules import Numerics


BRANCH_STREAMS = os.environ.get("CVFT_BRANCH_STREAMS", "1") != "0"      # LLM branch on a second stream (joint mode): 35.9 -> 30.1 ms/step


class JointLLMFlowModel(nn.Module):
    _side = None
    def __init__(self, llm: nn.Module, flow: nn.Module, training_mode: str = 'joint', llm_loss_weight: float = 1.0,
                 flow_loss_weight: float = 1.0, no_prompt_training: bool = True):
        super().__init__()
        assert training_mode in ('joint', 'llm_only', 'flow_only')
        self.llm, self.flow = llm, flow
        self.training_mode = training_mode
        self.llm_loss_weight, self.flow_loss_weight = llm_loss_weight, flow_loss_weight
        self.no_prompt_training = no_prompt_training
        self.mel_mean, self.mel_std = MEL_MEAN, MEL_STD

    def normalize_mel(self, mel: torch.Tensor) -> torch.Tensor:
        return (mel - self.mel_mean) / self.mel_std

    def set_numerics(self, num: Numerics):
        for m in (self.llm, self.flow):
            if hasattr(m, "numerics"):
                m.numerics = num

    def forward(self, batch: dict, device, draws: Optional[dict] = None) -> Dict[str, Any]:
        """llm_flow_model.py:77-107.  `draws` (optional) injects the CFM random draws."""
        losses: Dict[str, Any] = {}
        if self.training:
            HF.dropout_begin_step()       # new dropout masks per step (device-side seed: also across hipGraph replays)
        side = None
        if self.training_mode == 'joint' and BRANCH_STREAMS and torch.cuda.is_available():
            # The LLM and the Flow branch share nothing until the loss sum: run the LLM branch on a second stream so its
            # kernels overlap the estimator's under-filled ones (252-block GEMMs on 256 CUs).  autograd replays each
            # branch's backward on the stream its forward ran on, so the overlap holds for the whole step; inside a
            # captured hipGraph this is one fork and one join per direction.
            if JointLLMFlowModel._side is None:
                JointLLMFlowModel._side = torch.cuda.Stream()
            side = JointLLMFlowModel._side
            side.wait_stream(torch.cuda.current_stream())
        if self.training_mode in ('joint', 'llm_only'):
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                r = self._forward_llm(batch, device)
                losses['llm_loss'] = r['loss'] * self.llm_loss_weight
                if 'acc' in r:
                    losses['llm_acc'] = r['acc']
        if self.training_mode in ('joint', 'flow_only'):
            r = self._forward_flow(batch, device, draws)
            losses['flow_loss'] = r['loss'] * self.flow_loss_weight
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        if self.training_mode == 'joint':
            losses['loss'] = losses['llm_loss'] + losses['flow_loss']
        elif self.training_mode == 'llm_only':
            losses['loss'] = losses['llm_loss']
        else:
            losses['loss'] = losses['flow_loss']
        return losses

    def prepare_batch(self, batch: dict, device) -> dict:
        """Move a collated batch to `device` and attach the host-computed LLM index maps, so that the
        training step itself performs no host<->device transfers (hipGraph-capturable)."""
        out = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
        for k in ('speech_token_len', 'speech_feat_len', 'text_token_len'):
            if k in out:
                out[k] = out[k].to(torch.int32)
        if self.training_mode in ('joint', 'llm_only') and hasattr(self.llm, 'prepare_batch'):
            out.update(self.llm.prepare_batch(batch, device))
        return out

    def _forward_llm(self, batch: dict, device) -> Dict[str, Any]:
        return self.llm.forward_no_prompt(batch, device)

    def _forward_flow(self, batch: dict, device, draws=None) -> Dict[str, Any]:
        return self.flow.forward_no_prompt(batch, device, draws)


def build_joint_model(pretrained_path: str = PRETRAINED_MODEL_DIR, device: str = 'cuda', training_mode: str = 'joint',
                      llm_lora_config: Optional[dict] = None, flow_lora_config: Optional[dict] = None,
                      numerics: Optional[Numerics] = None) -> JointLLMFlowModel:
    """reference llm_flow_model.py:232-310.  The reference loads CosyVoice via hyperpyyaml; here the
    two CosyVoice-300M sub-models are constructed directly and ``llm.pt`` / ``flow.pt`` are loaded
    strictly when present (random init otherwise, e.g. for synthetic benchmarks)."""
    numerics = numerics or Numerics()
    print(f"[Joint] building CosyVoice-300M LLM + Flow (weights from {pretrained_path})")
    llm = build_llm_model(pretrained_path, device='cpu', numerics=numerics)
    flow = build_flow_model(pretrained_path, device='cpu', numerics=numerics)
    if training_mode in ('joint', 'llm_only') and llm_lora_config:
        st = apply_lora_to_model(llm, r=llm_lora_config.get('lora_r', 8), lora_alpha=llm_lora_config.get('lora_alpha', 16),
                                 lora_dropout=llm_lora_config.get('lora_dropout', 0.05),
                                 target_modules=llm_lora_config.get('target_modules', ['linear_q', 'linear_k', 'linear_v', 'linear_out', 'w_1', 'w_2']))
        print(f"  LLM LoRA: {st['replaced_layers']} layers, {st['trainable_params']:,} params ({st['trainable_ratio']:.2f}%)")
    if training_mode in ('joint', 'flow_only') and flow_lora_config:
        st = apply_lora_to_model(flow, r=flow_lora_config.get('lora_r', 16), lora_alpha=flow_lora_config.get('lora_alpha', 16),
                                 lora_dropout=flow_lora_config.get('lora_dropout', 0.05),
                                 target_modules=flow_lora_config.get('target_modules', ['to_q', 'to_k', 'to_v', 'linear_q', 'linear_k', 'linear_v', 'linear_out', 'w_1', 'w_2']))
        print(f"  Flow LoRA: {st['replaced_layers']} layers, {st['trainable_params']:,} params ({st['trainable_ratio']:.2f}%)")
    # freeze the branch the mode never trains (the reference leaves it
    # "trainable" but gradient-less; behaviourally identical, no 300M-param optimiser state).
    if training_mode == 'flow_only':
        llm.requires_grad_(False)
    if training_mode == 'llm_only':
        flow.requires_grad_(False)
    jc = JOINT_TRAINING_CONFIG
    model = JointLLMFlowModel(llm, flow, training_mode, jc.get('llm_loss_weight', 1.0), jc.get('flow_loss_weight', 1.0),
                              jc.get('no_prompt_training', True)).to(device)
    trainable = sum(p.numel() for p in model.parameters() if p.requires_grad)
    total = sum(p.numel() for p in model.parameters())
    print(f"[Joint] total {total:,} params, trainable {trainable:,} ({trainable / total * 100:.2f}%)")
    return model


def get_joint_merged_state_dict(model: JointLLMFlowModel) -> Dict[str, dict]:
    """reference llm_flow_model.py:313-336."""
    from .lora import get_merged_state_dict
    out = {}
    if any('lora_' in n for n, _ in model.llm.named_parameters()):
        out['llm'] = get_merged_state_dict(model.llm)
    if any('lora_' in n for n, _ in model.flow.named_parameters()):
        out['flow'] = get_merged_state_dict(model.flow)
    return out
