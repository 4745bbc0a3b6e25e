import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, contextlib
sys.argv = ["bench.py"]
import bench
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
from cosyvoice_lora_finetune_framework_amd.optim import FlatAdamW
from cosyvoice_lora_finetune_framework_amd.synthetic import synth_batch
dev = torch.device("cuda", 0)
jm = bench.build("joint", torch.bfloat16, dev, 16, 32)
opt = FlatAdamW([p for p in jm.parameters() if p.requires_grad], lr=2e-4)
batch = jm.prepare_batch(synth_batch([500] * 4, seed=1), dev)
def fb():
    out = jm(batch, dev)
    with HF.LoraGradSink():
        out['loss'].backward()
for _ in range(2):
    fb(); opt.zero_grad()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    fb()
    torch.cuda.synchronize()
import collections
c = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::add", "aten::add_", "aten::cat"):
        st = [s for s in (ev.stack or []) if "cosyvoice_lora" in s or "autograd" in s][:2]
        c[(ev.name, tuple(st))] += 1
for k, v in c.most_common(25):
    print(v, k)
