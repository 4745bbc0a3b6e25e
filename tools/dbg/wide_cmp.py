import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
from cosyvoice_lora_finetune_framework_amd.hipops.blockpack import BlockTailPack
from test_block_fused_gpu import _weights
M, DI, Fh, act = [int(x) if x.isdigit() else x for x in (sys.argv[1:5] if len(sys.argv) > 4 else ["37", "256", "128", "gelu_tanh"])]
w = _weights(DI, Fh, seed=M)
g = torch.Generator().manual_seed(M + 1)
o = torch.randn(M, DI, generator=g).to(torch.bfloat16)
x0 = (torch.randn(M, 256, generator=g) * 2.0 + 0.3).to(torch.bfloat16)
dy = torch.randn(M, 256, generator=g).to(torch.bfloat16)
wd = {k: v.cuda() for k, v in w.items()}
pack = BlockTailPack(wd["wo"], wd["bo"], wd["gamma"], wd["beta"], 1e-5, wd["w1"], wd["b1"], wd["w2"], wd["b2"])
res = {}
for lean in ("1", "2"):
    HF.BLOCK_LEAN = lean
    for rep in range(3):
        od, xd = o.cuda().requires_grad_(True), x0.cuda().requires_grad_(True)
        out = HF.block_tail(od, xd, pack, act)
        x1, z, mean, rstd = out.grad_fn.saved_tensors
        out.backward(dy.cuda())
        torch.cuda.synchronize()
        nt = -(-M // 32)
        cur = dict(out=out.detach().float(), x1=x1.float(), z=z.float()[:nt * 32 * Fh], mean=mean, rstd=rstd, dx=xd.grad.float(), do=od.grad.float())
        if lean in res:
            for k in cur:
                d = (cur[k] - res[lean][k]).abs().max().item()
                if d > 0:
                    print(f"lean={lean} rep {rep}: {k} differs from rep 0 by {d}")
        else:
            res[lean] = cur
for k in res["1"]:
    a, b = res["1"][k], res["2"][k]
    d = (a - b).abs()
    print(f"{k:5s}: max abs diff {d.max().item():.4g}  rel-L2 {((a - b).norm() / a.norm()).item():.3g}", end="")
    if d.max().item() > 0.05 and a.dim() == 2:
        bad = (d > 0.05).nonzero()
        print(f"   {len(bad)} bad; rows {sorted(set(bad[:, 0].tolist()))[:20]} cols {sorted(set(bad[:, 1].tolist()))[:40]}", end="")
    print()
for k in ("dx", "do"):
    d = (res["1"][k] - res["2"][k]).abs()
    R, Cc = d.shape
    print(k, "bad counts per [row tile][32-col tile]:")
    for rt in range(-(-R // 32)):
        print("   ", [int((d[32 * rt:32 * rt + 32, 32 * ct:32 * ct + 32] > 0.05).sum()) for ct in range(Cc // 32)])
