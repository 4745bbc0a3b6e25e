#!/usr/bin/env python3
"""Where do the torch-native kernels of one training step come from?  A TorchDispatchMode records every aten op that runs
during one eager forward + backward (and one optimiser step) with the innermost product-code frame that issued it
("<autograd engine>" when the C++ engine did: gradient accumulation of fan-outs, AccumulateGrad)."""
import collections
import contextlib
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.argv = ["bench.py"]
import bench
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
from cosyvoice_lora_finetune_framework_amd.optim import FlatAdamW
from cosyvoice_lora_finetune_framework_amd.synthetic import synth_batch

dev = torch.device("cuda", 0)
with contextlib.redirect_stdout(sys.stderr):
    jm = bench.build("joint", torch.bfloat16, dev, 16, 32, 1)
opt = FlatAdamW([p for p in jm.parameters() if p.requires_grad], lr=2e-4)
batch = jm.prepare_batch(synth_batch([500] * 16, seed=1), dev)


def fb():
    out = jm(batch, dev)
    with HF.LoraGradSink():
        out['loss'].backward()


for _ in range(2):
    fb()
    opt.zero_grad()
SKIP = ("aten.empty", "aten.view", "aten._unsafe_view", "aten.slice", "aten.select", "aten.as_strided", "aten.detach", "aten.t.",
        "aten.transpose", "aten.reshape", "aten.expand", "aten.unsqueeze", "aten.squeeze", "aten.alias", "aten.permute",
        "aten.split", "aten.unbind", "aten.empty_like", "aten.new_empty", "aten.empty_strided", "aten._local_scalar_dense",
        "aten.is_same_size", "aten.stride", "aten.size", "aten.sym_", "aten.lift_fresh", "aten.narrow", "aten.chunk")
cnt = collections.Counter()


class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            where = "<autograd engine>"
            for fr in reversed(traceback.extract_stack(limit=30)):
                if "cosyvoice_lora_finetune_framework_amd" in fr.filename:
                    where = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                    break
            cnt[(name, where)] += 1
        return func(*args, **(kwargs or {}))


with Rec():
    fb()
    opt.step(1.0)
    opt.zero_grad()
torch.cuda.synchronize()
tot = sum(cnt.values())
print(f"{tot} device-side aten ops in one step (views / allocations excluded)")
for (name, where), v in cnt.most_common(45):
    print(f"{v:5d}  {name:34s} {where}")
