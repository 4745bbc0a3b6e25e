"""Checkpoint -> merged weight files -- drop-in for the reference's merge_joint_weights.py
(``find_latest_joint_checkpoint`` :38-62, ``merge_llm_from_checkpoint`` :65-119, ``merge_flow_from_checkpoint``
:122-176, ``merge_both_from_checkpoint`` :179-273, CLI :276-356).  SURVEY 8f rank 1, second half.

Wire format in: a trainer checkpoint (``{'state_dict': {...}}`` or a bare state_dict) whose keys carry ``model.llm.`` /
``model.flow.`` (or ``llm.`` / ``flow.``) prefixes.  Wire format out: one torch ``state_dict`` per branch in the ORIGINAL
CosyVoice key names (LoRA folded into ``<path>.weight``), loadable with ``strict=True`` by the un-wrapped model.

Host-side only: no kernel runs here.  Unlike the reference the branch model is built once per branch (it rebuilds the
whole joint model for the flow half, :241-249, because merging mutates the weights in place -- the two branches share
nothing, so that rebuild is not needed) and callers may pass an already-built model (tests, tiny configs)."""
from __future__ import annotations

import argparse
import os
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from .config import JOINT_TRAINING_CONFIG, OUTPUT_DIR, PRETRAINED_MODEL_DIR
from .lora import get_merged_state_dict


def find_latest_joint_checkpoint(output_dir: str, mode: Optional[str] = None) -> Optional[str]:
    """Newest ``*.ckpt`` by mtime; `mode` filters on the file name, no mode prefers ``joint_joint`` files."""
    names = [f for f in os.listdir(output_dir) if f.endswith('.ckpt')]
    if mode:
        names = [f for f in names if mode in f]
    else:
        names = [f for f in names if 'joint_joint' in f] or names
    if not names:
        return None
    return os.path.join(output_dir, max(names, key=lambda f: os.path.getmtime(os.path.join(output_dir, f))))


def _read_state(ckpt_path: str) -> Dict[str, torch.Tensor]:
    ckpt = torch.load(ckpt_path, map_location='cpu', weights_only=True)     # tensors + plain scalars / strings only
    return ckpt.get('state_dict', ckpt)


def _fill_branch(branch: nn.Module, state: Dict[str, torch.Tensor], prefixes: Tuple[str, ...]) -> int:
    """Copy every checkpoint entry whose prefix-stripped key and shape match a branch entry; the rest of the branch
    keeps its (pretrained) values -- the reference's tolerant loader (:95-106)."""
    own = branch.state_dict()
    n = 0
    for key, value in state.items():
        name = next((key[len(p):] for p in prefixes if key.startswith(p)), key)
        if name in own and own[name].shape == value.shape:
            own[name] = value
            n += 1
    branch.load_state_dict(own)
    return n


def _build(mode: str):
    from .llm_flow_model import build_joint_model
    jc = JOINT_TRAINING_CONFIG
    return build_joint_model(pretrained_path=PRETRAINED_MODEL_DIR, device='cpu', training_mode=mode,
                             llm_lora_config=jc.get('llm_lora') if mode != 'flow_only' else None,
                             flow_lora_config=jc.get('flow_lora') if mode != 'llm_only' else None)


def _merge_branch(which: str, ckpt_path: str, output_path: str, model=None, state=None):
    print(f"[merge] {which}: checkpoint {ckpt_path}")
    state = _read_state(ckpt_path) if state is None else state
    model = model if model is not None else _build('llm_only' if which == 'llm' else 'flow_only')
    branch = getattr(model, which)
    n = _fill_branch(branch, state, (f'model.{which}.', f'{which}.'))
    print(f"[merge] {which}: {n} tensors taken from the checkpoint")
    merged = get_merged_state_dict(branch)
    torch.save(merged, output_path)
    print(f"[merge] {which}: wrote {output_path} ({os.path.getsize(output_path) / 2**20:.1f} MB)")
    return merged


def merge_llm_from_checkpoint(ckpt_path: str, output_path: str, model=None):
    return _merge_branch('llm', ckpt_path, output_path, model)


def merge_flow_from_checkpoint(ckpt_path: str, output_path: str, model=None):
    return _merge_branch('flow', ckpt_path, output_path, model)


def merge_both_from_checkpoint(ckpt_path: str, llm_output: str, flow_output: str, model=None):
    state = _read_state(ckpt_path)
    model = model if model is not None else _build('joint')
    return (_merge_branch('llm', ckpt_path, llm_output, model, state),
            _merge_branch('flow', ckpt_path, flow_output, model, state))


def main(argv=None):
    ap = argparse.ArgumentParser(description='fold the LoRA adapters of a joint-training checkpoint into llm / flow weight files')
    ap.add_argument('--ckpt', type=str)
    ap.add_argument('--llm-only', action='store_true')
    ap.add_argument('--flow-only', action='store_true')
    ap.add_argument('--llm-output', type=str, default=None)
    ap.add_argument('--flow-output', type=str, default=None)
    a = ap.parse_args(argv)
    llm_out = a.llm_output or os.path.join(OUTPUT_DIR, 'llm_merged.pt')
    flow_out = a.flow_output or os.path.join(OUTPUT_DIR, 'flow_merged.pt')
    ckpt = a.ckpt
    if ckpt and not os.path.exists(ckpt):
        print(f"error: checkpoint not found: {ckpt}")
        return 1
    if not ckpt:
        want = 'llm_only' if a.llm_only else 'flow_only' if a.flow_only else None
        ckpt = find_latest_joint_checkpoint(OUTPUT_DIR, want) if os.path.isdir(OUTPUT_DIR) else None
        if not ckpt and want:
            ckpt = find_latest_joint_checkpoint(OUTPUT_DIR, 'joint') if os.path.isdir(OUTPUT_DIR) else None
        if not ckpt:
            print(f"error: no .ckpt under {OUTPUT_DIR}")
            return 1
        print(f"[merge] latest checkpoint: {ckpt}")
    if a.llm_only:
        merge_llm_from_checkpoint(ckpt, llm_out)
    elif a.flow_only:
        merge_flow_from_checkpoint(ckpt, flow_out)
    else:
        merge_both_from_checkpoint(ckpt, llm_out, flow_out)
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
