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


# Sub-batches per branch, each an independent chain on its own stream.  Measured (B = 16, T = 500, captured step): the Flow
# branch in two halves next to the whole-batch LLM branch -- three chains -- 28.2 -> 27.0 ms; the estimator's kernels
# under-fill the chip even at M = 8 000 rows, and the second half fills what the first leaves.  More chains lose: LLM x 2 +
# Flow x 2 33.4 ms, Flow x 4 40.4 ms, Flow x 3 27.0 ms with 4 hardware queues but 37.3 ms with 8 -- every fork / join edge of
# the captured graph crosses hardware queues (~9 us each).  Results are unchanged (global loss denominators, see forward()).
# Single-branch modes: flow_only x 2 18.6 -> 18.1 ms (x 3 18.6), llm_only x 2 16.0 -> 15.3 ms.
SPLIT = {'llm': int(os.environ.get("CVFT_LLM_SPLIT", "1")), 'flow': int(os.environ.get("CVFT_FLOW_SPLIT", "2"))}      # joint / flow_only
SPLIT_LLM_ONLY = int(os.environ.get("CVFT_LLM_SPLIT", "2"))                                                             # llm_only
SPLIT_MIN_PART = int(os.environ.get("CVFT_SPLIT_MIN_PART", "4"))      # utterances per sub-batch below which a branch is not split
# flow_only has no LLM chain to overlap with: two Flow chains of 4 utterances are SLOWER than one of 8 (BASELINE configs[1], B = 8:
# 13.89 vs 13.56 ms/step, same-box A/B), two of 8 faster than one of 16 (18.1 vs 18.6)
SPLIT_MIN_PART_FLOW_ONLY = int(os.environ.get("CVFT_SPLIT_MIN_PART_FLOW_ONLY", "8"))
CHAINS_HINT = 0        # > 0: pin the concurrent-chains hint (bench.py's single-stream roofline leg times the kernels of the three-chain step)
BRANCH_STREAMS = os.environ.get("CVFT_BRANCH_STREAMS", "1") != "0"      # LLM branch on a second stream (joint mode): 35.9 -> 30.1 ms/step


LOSS_FUSE = os.environ.get("CVFT_LOSS_FUSE", "1") != "0"      # 0: the plain op-per-factor recombination (A/B)
# 1: the trainer issues every chain's backward right behind its forward (forward_backward: no join of all chains between the two
# directions); 0 (default): forward(), then ONE backward of the total, the reference's order.  Measured neutral, same box, 40 steps:
# joint 21.28 / 21.36 (1) vs 21.33 / 21.34 ms (0), flow_only 14.29 vs 14.27, llm_only 13.38 vs 13.31 -- the step is bound by the
# chip's CU-time, an idle chain is another chain's CUs
# Every chain's backward issued right behind its own forward on the chain's stream (forward_backward), and the chain's postponed
# adapter products flushed there too: "0" = the reference's order (forward of all chains, one weighted total, one backward).  Neutral
# while the optimiser ran outside the captured step (21.28 / 21.36 vs 21.33 / 21.34); with it inside, the step's single-stream tail
# behind the join is what the chains' own flush shortens: 20.13 / 20.20 / 20.19 -> 20.02 / 20.05 / 20.10 ms (three same-box pairs).
# "auto" (default): on in the joint mode; the single-branch modes lose 0.03 ms with it (flow_only 14.25 -> 14.28, llm_only 12.82 -> 12.86).
CHAIN_BWD = os.environ.get("CVFT_CHAIN_BWD", "auto")


def chain_bwd_on(model) -> bool:
    if CHAIN_BWD in ("0", "1"):
        return CHAIN_BWD == "1"
    return getattr(model, "training_mode", None) == 'joint'
CHAIN_FLUSH = os.environ.get("CVFT_CHAIN_FLUSH", "1") != "0"      # (with CHAIN_BWD: LoraGradSink.flush_chain behind each chain's backward)
# Diagnostic: with CVFT_CHAIN_EVENTS=1 (and CVFT_CHAIN_BWD=1) forward_backward drops a clock stamp (cvft_debug_stamp: a one-thread
# kernel node, HIP refuses timing events inside a captured graph) at the fork, behind every chain's forward and backward, and at the
# join; bench.py prints their offsets after the run: which chain ends the step, and how long the others have been done by then.
CHAIN_EVENTS = {} if os.environ.get("CVFT_CHAIN_EVENTS", "0") != "0" else None
_CHAIN_STAMPS = []


def _chain_event(key) -> None:
    if CHAIN_EVENTS is None:
        return
    if not _CHAIN_STAMPS:
        _CHAIN_STAMPS.append(torch.zeros(64, dtype=torch.int64, device='cuda'))
    slot = CHAIN_EVENTS.setdefault(key, len(CHAIN_EVENTS))
    from .hipops.binding import check, lib, ptr, stream
    check(lib().cvft_debug_stamp(ptr(_CHAIN_STAMPS[0]), slot, stream()), "cvft_debug_stamp")


def chain_event_offsets_ms():
    """{key: ms since the fork} of the last step that ran (diagnostic)"""
    if not CHAIN_EVENTS or not _CHAIN_STAMPS:
        return {}
    t = _CHAIN_STAMPS[0].cpu().tolist()
    t0 = t[CHAIN_EVENTS[('fork',)]]
    return {k: (t[s] - t0) / 1e5 for k, s in CHAIN_EVENTS.items()}
CHAIN_ORDER = int(os.environ.get("CVFT_CHAIN_ORDER", "0"))      # 1: chains enqueued Flow first, LLM last (the caller's stream then carries the LLM) -- see forward()
SIDE_STREAM_PRIORITY = int(os.environ.get("CVFT_SIDE_PRIO", "0"))      # priority of the side streams (the Flow chains in joint mode): -1 = high


def _scaled(x, w):
    """x * w without the launch when w is the Python number 1"""
    return x if (LOSS_FUSE and isinstance(w, (int, float)) and w == 1) else x * w


def _total(xs):
    """sum of 0-d tensors without `0 + x` (the built-in sum's start value costs a launch)"""
    if not LOSS_FUSE:
        return sum(xs)
    t = xs[0]
    for x in xs[1:]:
        t = t + x
    return t


class JointLLMFlowModel(nn.Module):
    _streams: list = []
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
        """llm_flow_model.py:77-107.  `draws` (optional) injects the CFM random draws.

        Execution (results unchanged): the LLM and the Flow branch share nothing until the loss sum, and utterances share
        nothing but the loss denominators -- so the step runs as SPLIT['llm'] + SPLIT['flow'] independent chains on separate HIP
        streams (LLM / Flow x sub-batches, `SPLIT`).  Most kernels of a chain under-fill the chip (the estimator's GEMMs are 252
        blocks on 256 CUs) and are latency-bound; concurrent chains fill it.  autograd replays every chain's backward on
        the stream its forward ran on, so the overlap holds for the whole step, and inside a captured hipGraph it costs
        one fork and one join per chain and direction.  Sub-batch losses are recombined with their share of the global
        denominators (frames for the CFM loss, target tokens for the CE loss): the reference's global-batch means."""
        losses: Dict[str, Any] = {}
        if self.training:
            HF.dropout_begin_step()       # new dropout masks per step (device-side seed: also across hipGraph replays)
        parts = batch.get('_parts')
        if parts is None:
            parts = self._make_parts(batch, batch, device, torch.cuda.is_available())
        do_llm = self.training_mode in ('joint', 'llm_only')
        do_flow = self.training_mode in ('joint', 'flow_only')
        chains = [('llm', k) for k in range(len(parts['llm']))] * do_llm + [('flow', k) for k in range(len(parts['flow']))] * do_flow
        if CHAIN_ORDER and len(chains) == 3:      # (A/B: which chain keeps the caller's stream and in which order the others fork)
            import itertools
            chains = [chains[i] for i in list(itertools.permutations(range(3)))[CHAIN_ORDER % 6]]
        use_streams = BRANCH_STREAMS and torch.cuda.is_available() and len(chains) > 1
        if torch.cuda.is_available():      # tile-shape hint for the kernels that would own a whole CU (cvft.h)
            HF.lib().cvft_set_concurrent_chains(CHAINS_HINT if CHAINS_HINT else (len(chains) if use_streams else 1))
        HF.LoraGradSink.uses_hint = max(len(v) for v in parts.values())      # chains that will run the same adapters
        cur = torch.cuda.current_stream() if use_streams else None
        results = {}
        for ci, (kind, k) in enumerate(reversed(chains)):          # the last chain (flow part 0 ... ) stays on the caller's stream
            st = None
            if use_streams and ci < len(chains) - 1:
                while len(JointLLMFlowModel._streams) <= ci:
                    JointLLMFlowModel._streams.append(torch.cuda.Stream(priority=SIDE_STREAM_PRIORITY))
                st = JointLLMFlowModel._streams[ci]
                st.wait_stream(cur)
            with (torch.cuda.stream(st) if st is not None else contextlib.nullcontext()):
                part = parts[kind][k]
                # (every scalar op here is a launch of its own on the joined stream, with the whole chip waiting between the
                # last forward chain and the first backward launch -- and another one or two in backward: unit factors are skipped)
                if kind == 'llm':
                    r = self._forward_llm(part, device)
                    results[(kind, k)] = (_scaled(r['loss'], part['_w_llm']), _scaled(r['acc'], part['_w_llm']) if 'acc' in r else None)
                else:
                    d = None if draws is None else {n_: v[part['_rows']] if '_rows' in part else v for n_, v in draws.items()}
                    r = self._forward_flow(part, device, d)
                    results[(kind, k)] = (_scaled(r['loss'], part['_w_flow']), None)
            if st is not None:
                results[(kind, k)] = (results[(kind, k)], st)
        for key, v in list(results.items()):                       # join
            if isinstance(v[0], tuple):
                cur.wait_stream(v[1])
                results[key] = v[0]
        if do_llm:
            losses['llm_loss'] = _scaled(_total([results[('llm', k)][0] for k in range(len(parts['llm']))]), self.llm_loss_weight)
            if results[('llm', 0)][1] is not None:
                losses['llm_acc'] = _total([results[('llm', k)][1] for k in range(len(parts['llm']))])
        if do_flow:
            losses['flow_loss'] = _scaled(_total([results[('flow', k)][0] for k in range(len(parts['flow']))]), self.flow_loss_weight)
        if self.training_mode == 'joint':
            losses['loss'] = losses['llm_loss'] + losses['flow_loss']
        elif self.training_mode == 'llm_only':
            losses['loss'] = losses['llm_loss']
        else:
            losses['loss'] = losses['flow_loss']
        return losses

    def forward_backward(self, batch: dict, device, draws: Optional[dict] = None, term_w=None, accum: int = 1) -> Dict[str, Any]:
        """forward() + the backward of  sum_k term_w[k] * losses[k + '_loss'] / accum  (k over the branches this mode runs, in the
        order 'llm', 'flow'; term_w: the trainer's data-parallel loss weights, a device vector, None = ones), CHAIN BY CHAIN:
        every chain's backward is issued on the chain's own stream right behind its forward.  The chains share nothing but leaf
        parameters and the loss denominators (host-known shares, `_w_llm` / `_w_flow`), so a chain's seed gradient --
        term_w[k] x branch weight x its share / accum -- is known before its forward has run; nothing makes the LLM chain's
        backward wait for the Flow chains' forward (forward() + loss.backward() joins all chains on one stream between the two
        directions: the earlier chain idles there and the chip runs under-filled until the last forward chain has finished).
        Call inside a LoraGradSink (the adapters' slab products of all chains meet in its one reduce).  Returns forward()'s
        dict, detached.  Same masks, same arithmetic per chain; the loss scalars are recombined exactly as in forward()."""
        losses: Dict[str, Any] = {}
        if self.training:
            HF.dropout_begin_step()
        parts = batch.get('_parts')
        if parts is None:
            parts = self._make_parts(batch, batch, device, torch.cuda.is_available())
        do_llm = self.training_mode in ('joint', 'llm_only')
        do_flow = self.training_mode in ('joint', 'flow_only')
        chains = [('llm', k) for k in range(len(parts['llm']))] * do_llm + [('flow', k) for k in range(len(parts['flow']))] * do_flow
        use_streams = BRANCH_STREAMS and torch.cuda.is_available() and len(chains) > 1
        if torch.cuda.is_available():
            HF.lib().cvft_set_concurrent_chains(CHAINS_HINT if CHAINS_HINT else (len(chains) if use_streams else 1))
        HF.LoraGradSink.uses_hint = max(len(v) for v in parts.values())
        cur = torch.cuda.current_stream() if use_streams else None
        # seed gradients, made on the caller's stream BEFORE the chains fork (every chain waits for that stream first)
        terms = [k for k in ('llm', 'flow') if (k == 'llm' and do_llm) or (k == 'flow' and do_flow)]
        branch_w = {'llm': float(self.llm_loss_weight), 'flow': float(self.flow_loss_weight)}
        seeds = {}
        for kind, k in chains:
            share = parts[kind][k]['_w_llm' if kind == 'llm' else '_w_flow']
            c = branch_w[kind] / float(accum)
            tw = None if term_w is None else term_w[terms.index(kind)]
            if torch.is_tensor(share):
                sd = share.to(torch.float32) * c
                sd = sd if tw is None else sd * tw
            elif tw is not None:
                sd = tw.to(torch.float32) * (c * float(share))
            else:
                sd = torch.full((), c * float(share), dtype=torch.float32, device=device)
            seeds[(kind, k)] = sd
        results = {}
        _chain_event(('fork',))
        for ci, (kind, k) in enumerate(reversed(chains)):
            st = None
            if use_streams and ci < len(chains) - 1:
                while len(JointLLMFlowModel._streams) <= ci:
                    JointLLMFlowModel._streams.append(torch.cuda.Stream())
                st = JointLLMFlowModel._streams[ci]
                st.wait_stream(cur)
            with (torch.cuda.stream(st) if st is not None else contextlib.nullcontext()):
                part = parts[kind][k]
                if kind == 'llm':
                    r = self._forward_llm(part, device)
                else:
                    d = None if draws is None else {n_: v[part['_rows']] if '_rows' in part else v for n_, v in draws.items()}
                    r = self._forward_flow(part, device, d)
                loss = r['loss']
                _chain_event((kind, k, 'forward done'))
                torch.autograd.backward([loss], [seeds[(kind, k)].to(loss.dtype)])
                if CHAIN_FLUSH and HF.LoraGradSink.active is not None:
                    HF.LoraGradSink.active.flush_chain()      # this chain's postponed adapter products, on its own stream
                _chain_event((kind, k, 'backward done'))
                share = part['_w_llm' if kind == 'llm' else '_w_flow']
                results[(kind, k)] = (_scaled(loss.detach(), share), _scaled(r['acc'].detach(), share) if 'acc' in r else None)
            if st is not None:
                results[(kind, k)] = (results[(kind, k)], st)
        for key, v in list(results.items()):                       # join
            if isinstance(v[0], tuple):
                cur.wait_stream(v[1])
                results[key] = v[0]
        _chain_event(('join',))
        if do_llm:
            losses['llm_loss'] = _scaled(_total([results[('llm', k)][0] for k in range(len(parts['llm']))]), self.llm_loss_weight)
            if results[('llm', 0)][1] is not None:
                losses['llm_acc'] = _total([results[('llm', k)][1] for k in range(len(parts['llm']))])
        if do_flow:
            losses['flow_loss'] = _scaled(_total([results[('flow', k)][0] for k in range(len(parts['flow']))]), self.flow_loss_weight)
        if self.training_mode == 'joint':
            losses['loss'] = losses['llm_loss'] + losses['flow_loss']
        elif self.training_mode == 'llm_only':
            losses['loss'] = losses['llm_loss']
        else:
            losses['loss'] = losses['flow_loss']
        return losses

    def _prepare_one(self, batch: dict, device, lm_pad: int = 1, with_llm: bool = True, lm_min: int = 0) -> dict:
        out = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
        for k in ('speech_token_len', 'speech_feat_len', 'text_token_len'):
            if k in out:
                out[k] = out[k].to(torch.int32)
        if with_llm and self.training_mode in ('joint', 'llm_only') and hasattr(self.llm, 'prepare_batch') and 'text_token' in batch:
            out.update(self.llm.prepare_batch(batch, device, lm_pad, lm_min))
        return out

    def _split_parts(self, batch: dict, device, nparts: int, kind: str = 'llm', lm_pad: int = 1, lm_min: int = 0):
        """Host side: contiguous sub-batches + their share of the global loss denominators (frames / target tokens).
        lm_min: LM length every sub-batch's index maps are padded to -- the captured layout's (Trainer._fit_layout matches
        layouts on the WHOLE batch's length; a sub-batch that derived its own shorter length would miss the captured step
        and be captured again)."""
        B = batch['speech_token'].shape[0]
        bounds = [round(i * B / nparts) for i in range(nparts + 1)]
        feat_len = batch['speech_feat_len'].detach().cpu().double()
        tgt_len = batch['speech_token_len'].detach().cpu().double() + 1.0          # speech tokens + EOS (llm.py:88-95 targets)
        parts = []
        for i in range(nparts):
            sl = slice(bounds[i], bounds[i + 1])
            sub = {k: (v[sl] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == B else v)
                   for k, v in batch.items() if not k.startswith('_')}
            if '_true_dims' in batch:                   # exact batch maxima of a shape-bucketed batch: whole-batch values
                sub['_true_dims'] = batch['_true_dims']
            part = self._prepare_one(sub, device, lm_pad, with_llm=(kind == 'llm'), lm_min=lm_min)   # index maps only where the LM runs
            part['_rows'] = sl
            # device scalars, not Python floats: a captured step is replayed on other batches (other length mixes)
            part['_w_flow'] = torch.tensor(float(feat_len[sl].sum() / feat_len.sum()), dtype=torch.float32).to(device)
            part['_w_llm'] = torch.tensor(float(tgt_len[sl].sum() / tgt_len.sum()), dtype=torch.float32).to(device)
            parts.append(part)
        return parts

    def prepare_batch(self, batch: dict, device, lm_pad: int = 1, lm_min: int = 0) -> dict:
        """Move a collated batch to `device` and attach the host-computed LLM index maps (and the sub-batch split), so
        that the training step itself performs no host<->device transfers (hipGraph-capturable)."""
        out = self._prepare_one(batch, device, lm_pad, lm_min=lm_min)
        out['_parts'] = self._make_parts(batch, out, device, True, lm_pad, lm_min)
        return out

    def _make_parts(self, batch: dict, whole: dict, device, split: bool, lm_pad: int = 1, lm_min: int = 0) -> dict:
        """per branch: the list of sub-batches its chains run on (one entry aliasing `whole` when the branch is not split)"""
        B = batch['speech_token'].shape[0]
        parts = {}
        for kind in ('llm', 'flow'):
            runs = self.training_mode in ('joint', f'{kind}_only')          # a branch the mode does not run is never split
            n = SPLIT_LLM_ONLY if (kind == 'llm' and self.training_mode == 'llm_only') else SPLIT[kind]
            min_part = SPLIT_MIN_PART_FLOW_ONLY if self.training_mode == 'flow_only' else SPLIT_MIN_PART
            if not (split and runs) or B < n * min_part:
                n = 1
            parts[kind] = self._split_parts(batch, device, n, kind, lm_pad, lm_min) if n > 1 else [dict(whole, _w_llm=1.0, _w_flow=1.0)]
        return parts

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
