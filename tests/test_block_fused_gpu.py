"""Row-tile chain kernels of the estimator transformer block (csrc/block_fused.hip) against
 (a) an fp64 torch restatement of the reference's op sequence (matcha transformer.py:290-316 == modules.py:362-375:
     to_out + residual, norm3, ff.net[0].proj + GELU, ff.net[2] + residual) on the same bf16-stored operands, and
 (b) the launch-per-stage form of the product path (GEMM launches + LayerNorm launch), forward and backward.
Tolerance: bf16 storage, relative L2 <= 2e-2 (same as tests/test_ops_gpu.py); the two product forms agree to 1e-2."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _weights(DI, Fh, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
    w = dict(wo=r(256, DI, sc=DI ** -0.5), bo=r(256, sc=0.1), gamma=1.0 + r(256, sc=0.2), beta=r(256, sc=0.1),
             w1=r(Fh, 256, sc=256 ** -0.5), b1=r(Fh, sc=0.1), w2=r(256, Fh, sc=Fh ** -0.5), b2=r(256, sc=0.1))
    # what the kernels see: bf16-stored weights, fp32 biases / affine
    for k in ("wo", "w1", "w2"):
        w[k] = w[k].to(torch.bfloat16).float()
    return w


def _ref(o, x0, w, act, with_o=True):
    """fp64 restatement; x1 is rounded to bf16 where the product path stores it (the LayerNorm reads the stored tensor)."""
    o, x0 = o.double().requires_grad_(True), x0.double().requires_grad_(True)
    W = {k: v.double() for k, v in w.items()}
    x1 = x0 + o @ W["wo"].t() + W["bo"] if with_o else x0
    x1r = x1 + (x1.detach().to(torch.bfloat16).double() - x1.detach())          # straight-through rounding
    y = F.layer_norm(x1r, (256,), W["gamma"], W["beta"], 1e-5)
    z = y @ W["w1"].t() + W["b1"]
    h = F.gelu(z, approximate="tanh" if act == "gelu_tanh" else "none")
    out = x1r + h @ W["w2"].t() + W["b2"]
    return o, x0, x1r, out


@pytest.mark.parametrize("act", ["gelu_erf", "gelu_tanh"])
@pytest.mark.parametrize("M,DI,Fh,with_o", [(64, 512, 1024, True), (250, 512, 1024, True), (37, 256, 128, True), (70, 512, 256, True), (96, 512, 1024, False),
                                            (4000, 512, 1024, True)])
def test_block_tail_matches_fp64_reference(act, M, DI, Fh, with_o):
    from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
    from cosyvoice_lora_finetune_framework_amd.hipops.blockpack import BlockTailPack
    w = _weights(DI, Fh, seed=M)
    g = torch.Generator().manual_seed(M + 1)
    o = (torch.randn(M, DI, generator=g)).to(torch.bfloat16)
    x0 = (torch.randn(M, 256, generator=g) * 2.0 + 0.3).to(torch.bfloat16)
    dy = (torch.randn(M, 256, generator=g)).to(torch.bfloat16)
    wd = {k: v.to(DEV) for k, v in w.items()}
    pack = BlockTailPack(wd["wo"] if with_o else None, wd["bo"], wd["gamma"], wd["beta"], 1e-5, wd["w1"], wd["b1"], wd["w2"], wd["b2"])
    od = o.to(DEV).requires_grad_(True)
    xd = x0.to(DEV).requires_grad_(True)
    out = HF.block_tail(od if with_o else None, xd, pack, act)
    out.backward(dy.to(DEV))
    ro, rx, _, rout = _ref(o.float(), x0.float(), w, act, with_o)
    rout.backward(dy.double())
    assert rel(out, rout) < 2e-2, rel(out, rout)
    assert rel(xd.grad, rx.grad) < 2e-2, rel(xd.grad, rx.grad)
    if with_o:
        assert rel(od.grad, ro.grad) < 2e-2, rel(od.grad, ro.grad)
    # exact-integer style layout check: a wrong fragment map shows up as O(1) error, far above bf16 rounding
    assert float((out.float().cpu() - rout.float()).abs().max()) < 0.25


def test_block_tail_equals_launch_per_stage_form():
    """The same half block through the product path's separate launches (GEMM + LayerNorm + fused feed-forward Function)."""
    import torch.nn as nn
    from cosyvoice_lora_finetune_framework_amd import modules as Mo
    from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
    torch.manual_seed(3)
    blk = Mo.BasicTransformerBlock(256, 8, 64, 0.0, "gelu").to(DEV)
    for p in blk.parameters():
        p.requires_grad_(False)
        p.copy_(p.to(torch.bfloat16).float())
    nn.init.normal_(blk.attn1.to_out[0].bias, std=0.1)
    nn.init.normal_(blk.norm3.bias, std=0.1)
    M = 500
    g = torch.Generator().manual_seed(5)
    o = torch.randn(M, 512, generator=g).to(torch.bfloat16).to(DEV)
    x0 = torch.randn(M, 256, generator=g).to(torch.bfloat16).to(DEV)
    dy = torch.randn(M, 256, generator=g).to(torch.bfloat16).to(DEV)
    res = {}
    for fuse in (False, True):
        HF.BLOCK_FUSE = fuse
        try:
            oo, xx = o.clone().requires_grad_(True), x0.clone().requires_grad_(True)
            out = blk._tail(oo, xx, "gelu_erf")
            out.backward(dy)
            res[fuse] = (out.detach(), oo.grad, xx.grad)
        finally:
            HF.BLOCK_FUSE = True
    for a, b in zip(res[True], res[False]):
        assert rel(a, b) < 1e-2, rel(a, b)
