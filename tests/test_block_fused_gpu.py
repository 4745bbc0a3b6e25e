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


@pytest.mark.parametrize("lean", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("act", ["gelu_erf", "gelu_tanh"])
@pytest.mark.parametrize("M,DI,Fh,with_o", [(64, 512, 1024, True), (250, 512, 1024, True), (37, 256, 128, True), (70, 512, 256, True), (96, 512, 1024, False),
                                            (4000, 512, 1024, True)])
def test_block_tail_matches_fp64_reference(act, M, DI, Fh, with_o, lean, monkeypatch):
    """lean = the CU-sharing form of the kernels (csrc/block_lean.hip: each wave owns 64 output features, hidden tiles exchanged
    through LDS, <= 256 registers); both forms against the same fp64 restatement."""
    from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
    monkeypatch.setattr(HF, "BLOCK_LEAN", str(lean))
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


def _qkv_case(M, p, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
    bq = lambda t: t.to(torch.bfloat16).float()
    w = dict(wqkv=bq(r(1536, 256, sc=256 ** -0.5)), bias=r(1536, sc=0.1), gamma=1.0 + r(256, sc=0.2), beta=r(256, sc=0.1),
             A=[bq(r(16, 256, sc=0.2)) for _ in range(3)], B=[bq(r(512, 16, sc=0.2)) for _ in range(3)])
    x = bq(r(M, 256) * 1.5 + 0.2)
    dY = bq(r(M, 1536))
    dres = bq(r(M, 256))
    return w, x, dY, dres


def _keep_mask(M, Kd, p, seed, site):
    """host replica of the counter-based mask (csrc/common.h cvft_keep4; tests/helpers.py keep_fields_host): 1.0 where kept"""
    from helpers import drop_thr_host, keep_fields_host
    fields = keep_fields_host(int(seed), int(site), M * Kd // 4).reshape(M, Kd)
    return torch.from_numpy((fields >= drop_thr_host(p)).astype("float32"))


@pytest.mark.parametrize("wide", [0, 1])                                    # (1: 64 rows per workgroup, csrc/block_qkv_wide.hip)
@pytest.mark.parametrize("M,p", [(64, 0.0), (250, 0.05), (37, 0.3), (2000, 0.05), (4200, 0.05), (4200, 0.0)])      # (> 128 row tiles: one workgroup per tile)
def test_block_qkv_matches_fp64_reference(M, p, wide):
    """norm1 + stacked LoRA q|k|v with lora_dropout (lora.py:64-76 x 3 over matcha transformer.py:255-262) and its backward, fused
    launches against an fp64 restatement under host-replicated masks; also V, U and the dropped copies the adapter gradients use."""
    from cosyvoice_lora_finetune_framework_amd.hipops import binding as cb
    from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
    from cosyvoice_lora_finetune_framework_amd.hipops.blockpack import BlockQkvPack
    import ctypes as C
    w, x, dY, dres = _qkv_case(M, p, seed=M)
    scale = 2.0
    d = lambda t: t.to(DEV)
    pack = BlockQkvPack(d(w["wqkv"]), d(w["bias"]), d(w["gamma"]), d(w["beta"]), 1e-5)
    A = torch.cat(w["A"], 0).to(torch.bfloat16)                        # [48, 256]
    Bb = torch.zeros(1536, 48)
    for t in range(3):
        Bb[512 * t:512 * (t + 1), 16 * t:16 * (t + 1)] = w["B"][t]
    Bb = Bb.to(torch.bfloat16)
    ops = (d(A), d(A.t().contiguous()), d(Bb), d(Bb.t().contiguous()))
    seed_val, sites = 123456789, [5, 9, 11]
    HF._DROPOUT["seed"] = torch.full((1,), seed_val, dtype=torch.int64, device=DEV)
    xd_ = d(x.to(torch.bfloat16))
    Y = torch.empty((M, 1536), dtype=torch.bfloat16, device=DEV)
    U = torch.empty((M, 48), dtype=torch.bfloat16, device=DEV)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    outs = [torch.empty((M, 256), dtype=torch.bfloat16, device=DEV) for _ in range(3)]
    a = cb.BlockQkvArgs()
    a.M, a.x, a.gamma, a.beta, a.eps, a.mean, a.rstd = M, cb.ptr(xd_), cb.ptr(pack.gamma), cb.ptr(pack.beta), 1e-5, cb.ptr(mean), cb.ptr(rstd)
    a.W_fwd, a.bias, a.N3, a.wide = cb.ptr(pack.W_fwd), cb.ptr(pack.bias), 1536, wide
    a.A, a.lda, a.Bb, a.ldb = cb.ptr(ops[0]), 256, cb.ptr(ops[2]), 48
    a.alpha, a.p = scale, p
    if p > 0:
        a.seed = cb.ptr(HF._DROPOUT["seed"])
        for i in range(3):
            a.sites[i] = sites[i]
            a.xd[i] = outs[i].data_ptr()
    else:
        a.y_out = cb.ptr(outs[0])
    a.U, a.ldu, a.Y, a.ldy = cb.ptr(U), 48, cb.ptr(Y), 1536
    cb.check(cb.lib().cvft_block_qkv_fwd(C.byref(a), cb.stream()), "fwd")
    # ---- fp64 reference (y rounded to bf16 where the kernel rounds it; U rounded where it is stored)
    xr = x.double()
    y = F.layer_norm(xr, (256,), w["gamma"].double(), w["beta"].double(), 1e-5).to(torch.bfloat16).double()
    keep = [(_keep_mask(M, 256, p, seed_val, sites[t]).double() if p > 0 else torch.ones(M, 256, dtype=torch.float64)) for t in range(3)]
    inv = 1.0 / (1.0 - p)
    Uref = torch.cat([scale * inv * ((keep[t] * y) @ w["A"][t].double().t()) for t in range(3)], 1)
    Yref = y @ w["wqkv"].double().t() + w["bias"].double() + Uref.to(torch.bfloat16).double() @ Bb.double().t()
    assert rel(U, Uref) < 1e-2 and rel(Y, Yref) < 1e-2, (rel(U, Uref), rel(Y, Yref))
    assert rel(mean, xr.mean(1)) < 1e-5
    if p > 0:
        for t in range(3):
            assert rel(outs[t], keep[t] * y * inv) < 5e-3
    else:
        assert rel(outs[0], y) < 5e-3
    # ---- backward
    V = torch.empty((M, 48), dtype=torch.bfloat16, device=DEV)
    dx = torch.empty((M, 256), dtype=torch.bfloat16, device=DEV)
    dYd, dresd = d(dY.to(torch.bfloat16)), d(dres.to(torch.bfloat16))
    b = cb.BlockQkvBwdArgs()
    b.M, b.dY, b.lddy, b.dres, b.x = M, cb.ptr(dYd), 1536, cb.ptr(dresd), cb.ptr(xd_)
    b.gamma, b.mean, b.rstd, b.W_bwd, b.N3 = cb.ptr(pack.gamma), cb.ptr(mean), cb.ptr(rstd), cb.ptr(pack.W_bwd_wide if wide else pack.W_bwd), 1536
    b.wide = wide
    b.At, b.ldat, b.Bbt, b.ldbt = cb.ptr(ops[1]), 48, cb.ptr(ops[3]), 1536
    b.alpha, b.p = scale, p
    if p > 0:
        b.seed = cb.ptr(HF._DROPOUT["seed"])
        for i in range(3):
            b.sites[i] = sites[i]
    b.V, b.ldv, b.dx = cb.ptr(V), 48, cb.ptr(dx)
    cb.check(cb.lib().cvft_block_qkv_bwd(C.byref(b), cb.stream()), "bwd")
    Vref = scale * dY.double() @ Bb.double()
    dy = dY.double() @ w["wqkv"].double()
    Vb = Vref.to(torch.bfloat16).double()
    for t in range(3):
        dy = dy + keep[t] * inv * (Vb[:, 16 * t:16 * (t + 1)] @ w["A"][t].double())
    xq = xr.clone().requires_grad_(True)
    yq = F.layer_norm(xq, (256,), w["gamma"].double(), w["beta"].double(), 1e-5)
    (yq * dy).sum().backward()
    dxref = dres.double() + xq.grad
    assert rel(V, Vref) < 1e-2, rel(V, Vref)
    assert rel(dx, dxref) < 1.5e-2, rel(dx, dxref)
    HF._DROPOUT["seed"] = None


def _tail_args(cb, M, o, x0, pack, x1, out, mean, rstd, z, act):
    a = cb.BlockTailArgs()
    a.M, a.o, a.ldo, a.x0, a.DI, a.W_fwd, a.bo = M, cb.ptr(o), o.stride(0), cb.ptr(x0), pack.DI, cb.ptr(pack.W_fwd), cb.ptr(pack.bo)
    a.x1, a.gamma, a.beta, a.eps = cb.ptr(x1), cb.ptr(pack.gamma), cb.ptr(pack.beta), pack.eps
    a.b1, a.F, a.b2, a.act = cb.ptr(pack.b1), pack.F, cb.ptr(pack.b2), act
    a.z, a.mean, a.rstd, a.out, a.lean = cb.ptr(z), cb.ptr(mean), cb.ptr(rstd), cb.ptr(out), 0
    return a


@pytest.mark.parametrize("act", ["gelu_erf", "gelu_tanh"])
@pytest.mark.parametrize("M,Fh,p", [(64, 1024, 0.05), (250, 1024, 0.0), (37, 256, 0.3), (4000, 1024, 0.05), (2000, 512, 0.0)])
def test_block_link_equals_tail_then_head_bitwise(M, Fh, p, act):
    """cvft_block_link_fwd (block i's tail + block i + 1's q|k|v head in one launch, the block output kept in registers) against the
    two launches it replaces, cvft_block_tail_fwd then cvft_block_qkv_fwd on the stored output: every output of both -- x1, out,
    the saved gelu' workspace, both LayerNorms' statistics, U, Y, the dropped copies -- bit for bit (same arithmetic in the same
    order, same masks); ragged M (rows beyond M untouched)."""
    from cosyvoice_lora_finetune_framework_amd.hipops import binding as cb
    from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
    from cosyvoice_lora_finetune_framework_amd.hipops.blockpack import BlockLinkPack, BlockQkvPack, BlockTailPack
    import ctypes as C
    d = lambda t: t.to(DEV)
    w = _weights(512, Fh, seed=M)
    wd = {k: d(v) for k, v in w.items()}
    tpack = BlockTailPack(wd["wo"], wd["bo"], wd["gamma"], wd["beta"], 1e-5, wd["w1"], wd["b1"], wd["w2"], wd["b2"])
    wq, _, _, _ = _qkv_case(M, p, seed=M + 7)
    hpack = BlockQkvPack(d(wq["wqkv"]), d(wq["bias"]), d(wq["gamma"]), d(wq["beta"]), 1e-5)
    lpack = BlockLinkPack(tpack, hpack)
    A = torch.cat(wq["A"], 0).to(torch.bfloat16)
    Bb = torch.zeros(1536, 48)
    for t in range(3):
        Bb[512 * t:512 * (t + 1), 16 * t:16 * (t + 1)] = wq["B"][t]
    A, Bb = d(A), d(Bb.to(torch.bfloat16))
    g = torch.Generator().manual_seed(M + 1)
    o = d(torch.randn(M, 512, generator=g).to(torch.bfloat16))
    x0 = d((torch.randn(M, 256, generator=g) * 2.0 + 0.3).to(torch.bfloat16))
    HF._DROPOUT["seed"] = torch.full((1,), 987654321, dtype=torch.int64, device=DEV)
    sites = [3, 4, 8]
    zrows = -(-M // 64) * 64
    res = {}
    for linked in (False, True):
        bf = lambda *s: torch.full(s, 7.0, dtype=torch.bfloat16, device=DEV)       # (sentinel: untouched tails must match too)
        x1, out, z = bf(M, 256), bf(M, 256), bf(zrows * Fh)
        mean, rstd, mean2, rstd2 = (torch.full((M,), 7.0, device=DEV) for _ in range(4))
        Y, U = bf(M, 1536), bf(M, 48)
        outs = [bf(M, 256) for _ in range(3)]
        a = _tail_args(cb, M, o, x0, tpack, x1, out, mean, rstd, z, HF.ACT[act])
        q = cb.BlockQkvArgs()
        q.M, q.x, q.gamma, q.beta, q.eps, q.mean, q.rstd = M, cb.ptr(out), cb.ptr(hpack.gamma), cb.ptr(hpack.beta), 1e-5, cb.ptr(mean2), cb.ptr(rstd2)
        q.W_fwd, q.bias, q.N3, q.wide = cb.ptr(hpack.W_fwd), cb.ptr(hpack.bias), 1536, 0
        q.A, q.lda, q.Bb, q.ldb = cb.ptr(A), 256, cb.ptr(Bb), 48
        q.alpha, q.p = 2.0, p
        if p > 0:
            q.seed = cb.ptr(HF._DROPOUT["seed"])
            for i in range(3):
                q.sites[i] = sites[i]
                q.xd[i] = outs[i].data_ptr()
        else:
            q.y_out = cb.ptr(outs[0])
        q.U, q.ldu, q.Y, q.ldy = cb.ptr(U), 48, cb.ptr(Y), 1536
        if linked:
            cb.check(cb.lib().cvft_block_link_fwd(C.byref(a), C.byref(q), cb.ptr(lpack.W_fwd), cb.stream()), "link")
        else:
            cb.check(cb.lib().cvft_block_tail_fwd(C.byref(a), cb.stream()), "tail")
            cb.check(cb.lib().cvft_block_qkv_fwd(C.byref(q), cb.stream()), "head")
        torch.cuda.synchronize()
        res[linked] = dict(x1=x1, out=out, z=z[:M * Fh] if M % 32 == 0 else z, mean=mean, rstd=rstd, mean2=mean2, rstd2=rstd2, Y=Y, U=U,
                           **{f"xd{i}": t for i, t in enumerate(outs)})
    HF._DROPOUT["seed"] = None
    for k in res[True]:
        assert torch.equal(res[True][k], res[False][k]), (k, rel(res[True][k], res[False][k]))
    assert float(res[True]["Y"].float().abs().max()) > 0.1


def test_block_link_refuses_other_forms():
    """the linked launch exists for the 32-row forms with the output projection only: anything else is an argument error, not a
    silently different kernel"""
    from cosyvoice_lora_finetune_framework_amd.hipops import binding as cb
    import ctypes as C
    a, q = cb.BlockTailArgs(), cb.BlockQkvArgs()
    a.M = q.M = 64
    a.F, q.N3, a.DI, a.lean = 1024, 1536, 512, 2
    w = torch.zeros(16, device=DEV)
    assert cb.lib().cvft_block_link_fwd(C.byref(a), C.byref(q), cb.ptr(w), cb.stream()) != 0


@pytest.mark.parametrize("act", ["gelu_erf", "gelu_tanh"])
@pytest.mark.parametrize("M,Fh,p", [(64, 1024, 0.05), (250, 1024, 0.0), (37, 256, 0.3), (4000, 1024, 0.05), (2000, 512, 0.0)])
def test_block_link_bwd_equals_head_then_tail_bitwise(M, Fh, p, act):
    """cvft_block_link_bwd (block i + 1's head backward + block i's tail backward in one launch, the boundary gradient kept in
    registers) against cvft_block_qkv_bwd then cvft_block_tail_bwd on the stored gradient: V, dx (still written), dx1 and dout bit
    for bit; ragged M."""
    from cosyvoice_lora_finetune_framework_amd.hipops import binding as cb
    from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
    from cosyvoice_lora_finetune_framework_amd.hipops.blockpack import BlockLinkPack, BlockQkvPack, BlockTailPack
    import ctypes as C
    d = lambda t: t.to(DEV)
    w = _weights(512, Fh, seed=M)
    wd = {k: d(v) for k, v in w.items()}
    tpack = BlockTailPack(wd["wo"], wd["bo"], wd["gamma"], wd["beta"], 1e-5, wd["w1"], wd["b1"], wd["w2"], wd["b2"])
    wq, x, dY, dres = _qkv_case(M, p, seed=M + 7)
    hpack = BlockQkvPack(d(wq["wqkv"]), d(wq["bias"]), d(wq["gamma"]), d(wq["beta"]), 1e-5)
    lpack = BlockLinkPack(tpack, hpack)
    A = torch.cat(wq["A"], 0).to(torch.bfloat16)
    Bb = torch.zeros(1536, 48)
    for t in range(3):
        Bb[512 * t:512 * (t + 1), 16 * t:16 * (t + 1)] = wq["B"][t]
    At, Bbt = d(A.t().contiguous()), d(Bb.to(torch.bfloat16).t().contiguous())
    g = torch.Generator().manual_seed(M + 1)
    o = d(torch.randn(M, 512, generator=g).to(torch.bfloat16))
    x0 = d((torch.randn(M, 256, generator=g) * 2.0 + 0.3).to(torch.bfloat16))
    HF._DROPOUT["seed"] = torch.full((1,), 987654321, dtype=torch.int64, device=DEV)
    sites = [3, 4, 8]
    # forward of the tail (its saved tensors) and the head's statistics on the tail's output
    bfz = lambda *s: torch.zeros(s, dtype=torch.bfloat16, device=DEV)
    x1, out, z = bfz(M, 256), bfz(M, 256), bfz(-(-M // 64) * 64 * Fh)
    mean, rstd = torch.zeros(M, device=DEV), torch.zeros(M, device=DEV)
    a = _tail_args(cb, M, o, x0, tpack, x1, out, mean, rstd, z, HF.ACT[act])
    cb.check(cb.lib().cvft_block_tail_fwd(C.byref(a), cb.stream()), "tail fwd")
    of = out.float()
    mean2 = of.mean(1).contiguous()
    rstd2 = (of.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    dYd, dresd = d(dY.to(torch.bfloat16)), d(dres.to(torch.bfloat16))
    res = {}
    for linked in (False, True):
        bf = lambda *s: torch.full(s, 7.0, dtype=torch.bfloat16, device=DEV)
        V, dx, dx1, do = bf(M, 48), bf(M, 256), bf(M, 256), bf(M, 512)
        b = cb.BlockQkvBwdArgs()
        b.M, b.dY, b.lddy, b.dres, b.x = M, cb.ptr(dYd), 1536, cb.ptr(dresd), cb.ptr(out)
        b.gamma, b.mean, b.rstd, b.W_bwd, b.N3, b.wide = cb.ptr(hpack.gamma), cb.ptr(mean2), cb.ptr(rstd2), cb.ptr(hpack.W_bwd), 1536, 0
        b.At, b.ldat, b.Bbt, b.ldbt = cb.ptr(At), 48, cb.ptr(Bbt), 1536
        b.alpha, b.p = 2.0, p
        if p > 0:
            b.seed = cb.ptr(HF._DROPOUT["seed"])
            for i in range(3):
                b.sites[i] = sites[i]
        b.V, b.ldv, b.dx = cb.ptr(V), 48, cb.ptr(dx)
        t = cb.BlockTailBwdArgs()
        t.M, t.x1, t.dy, t.gamma, t.mean, t.rstd, t.z = M, cb.ptr(x1), cb.ptr(dx), cb.ptr(tpack.gamma), cb.ptr(mean), cb.ptr(rstd), cb.ptr(z)
        t.W_bwd, t.F, t.DI, t.act, t.dx1, t.lean = cb.ptr(tpack.W_bwd), Fh, 512, HF.ACT[act], cb.ptr(dx1), 0
        t.dout, t.lddo = cb.ptr(do), 512
        if linked:
            cb.check(cb.lib().cvft_block_link_bwd(C.byref(b), C.byref(t), cb.ptr(lpack.W_bwd), cb.stream()), "link bwd")
        else:
            cb.check(cb.lib().cvft_block_qkv_bwd(C.byref(b), cb.stream()), "head bwd")
            cb.check(cb.lib().cvft_block_tail_bwd(C.byref(t), cb.stream()), "tail bwd")
        torch.cuda.synchronize()
        res[linked] = dict(V=V, dx=dx, dx1=dx1, do=do)
    HF._DROPOUT["seed"] = None
    for k in res[True]:
        assert torch.equal(res[True][k], res[False][k]), (k, rel(res[True][k], res[False][k]))
    assert float(res[True]["do"].float().abs().max()) > 1e-3 and torch.isfinite(res[True]["do"].float()).all()


@pytest.mark.parametrize("form", ["tail32", "tail64", "link"])
@pytest.mark.parametrize("M,T,with_lo", [(64, 32, True), (250, 125, True), (4000, 500, False), (2000, 250, True)])
def test_tail_backward_delta_for_the_attention_backward(M, T, with_lo, form):
    """cvft_block_tail_bwd / cvft_block_link_bwd also form delta[b, h, t] = sum over head h's 64 columns of dout . (attn_o + attn_o_lo)
    -- what the attention backward that consumes dout needs of its own output (it is then called with o == NULL) -- from the dout
    tile they are about to store: against the same sum taken in fp64 from the STORED dout, every form that offers it (32-row,
    64-row, the linked launch), ragged row tiles, with and without the residual; the other outputs do not change."""
    from cosyvoice_lora_finetune_framework_amd.hipops import binding as cb
    from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
    from cosyvoice_lora_finetune_framework_amd.hipops.blockpack import BlockLinkPack, BlockQkvPack, BlockTailPack
    import ctypes as C
    d = lambda t: t.to(DEV)
    Fh, B = 1024, M // T
    w = _weights(512, Fh, seed=M)
    wd = {k: d(v) for k, v in w.items()}
    tpack = BlockTailPack(wd["wo"], wd["bo"], wd["gamma"], wd["beta"], 1e-5, wd["w1"], wd["b1"], wd["w2"], wd["b2"])
    wq, x, dY, dres = _qkv_case(M, 0.0, seed=M + 7)
    hpack = BlockQkvPack(d(wq["wqkv"]), d(wq["bias"]), d(wq["gamma"]), d(wq["beta"]), 1e-5)
    lpack = BlockLinkPack(tpack, hpack)
    A = torch.cat(wq["A"], 0).to(torch.bfloat16)
    Bb = torch.zeros(1536, 48)
    for t in range(3):
        Bb[512 * t:512 * (t + 1), 16 * t:16 * (t + 1)] = wq["B"][t]
    At, Bbt = d(A.t().contiguous()), d(Bb.to(torch.bfloat16).t().contiguous())
    g = torch.Generator().manual_seed(M + 1)
    o = d(torch.randn(M, 512, generator=g).to(torch.bfloat16))
    o_lo = d((torch.randn(M, 512, generator=g) * 2 ** -9).to(torch.bfloat16)) if with_lo else None
    x0 = d((torch.randn(M, 256, generator=g) * 2.0 + 0.3).to(torch.bfloat16))
    dy = d(torch.randn(M, 256, generator=g).to(torch.bfloat16))
    bfz = lambda *s: torch.zeros(s, dtype=torch.bfloat16, device=DEV)
    x1, out, z = bfz(M, 256), bfz(M, 256), bfz(-(-M // 64) * 64 * Fh)
    mean, rstd = torch.zeros(M, device=DEV), torch.zeros(M, device=DEV)
    a = _tail_args(cb, M, o, x0, tpack, x1, out, mean, rstd, z, HF.ACT["gelu_erf"])
    cb.check(cb.lib().cvft_block_tail_fwd(C.byref(a), cb.stream()), "tail fwd")
    of = out.float()
    mean2, rstd2 = of.mean(1).contiguous(), (of.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    dYd, dresd = d(dY.to(torch.bfloat16)), d(dres.to(torch.bfloat16))
    res = {}
    for want in (False, True):
        V, dx, dx1, do = bfz(M, 48), bfz(M, 256), bfz(M, 256), bfz(M, 512)
        delta = torch.full((B, 8, T), 7.0, device=DEV)
        t = cb.BlockTailBwdArgs()
        t.M, t.x1, t.gamma, t.mean, t.rstd, t.z = M, cb.ptr(x1), cb.ptr(tpack.gamma), cb.ptr(mean), cb.ptr(rstd), cb.ptr(z)
        t.F, t.DI, t.act, t.dx1, t.dout, t.lddo = Fh, 512, HF.ACT["gelu_erf"], cb.ptr(dx1), cb.ptr(do), 512
        if want:
            t.attn_o, t.attn_o_lo, t.ldao, t.delta, t.T = cb.ptr(o), cb.ptr(o_lo), 512, cb.ptr(delta), T
        if form == "link":
            b = cb.BlockQkvBwdArgs()
            b.M, b.dY, b.lddy, b.dres, b.x = M, cb.ptr(dYd), 1536, cb.ptr(dresd), cb.ptr(out)
            b.gamma, b.mean, b.rstd, b.W_bwd, b.N3, b.wide = cb.ptr(hpack.gamma), cb.ptr(mean2), cb.ptr(rstd2), cb.ptr(hpack.W_bwd), 1536, 0
            b.At, b.ldat, b.Bbt, b.ldbt, b.alpha, b.p = cb.ptr(At), 48, cb.ptr(Bbt), 1536, 2.0, 0.0
            b.V, b.ldv, b.dx = cb.ptr(V), 48, cb.ptr(dx)
            t.dy, t.W_bwd, t.lean = cb.ptr(dx), cb.ptr(tpack.W_bwd), 0
            keep = (b,)
            cb.check(cb.lib().cvft_block_link_bwd(C.byref(b), C.byref(t), cb.ptr(lpack.W_bwd), cb.stream()), "link bwd")
        else:
            lean = 0 if form == "tail32" else 2
            t.dy, t.W_bwd, t.lean = cb.ptr(dy), cb.ptr(tpack.W_bwd if lean == 0 else tpack.W_bwd_wide), lean
            cb.check(cb.lib().cvft_block_tail_bwd(C.byref(t), cb.stream()), "tail bwd")
        torch.cuda.synchronize()
        res[want] = (dx1, do, delta)
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])      # asking for delta changes nothing else
    assert float(res[False][2].min()) == 7.0                                                          # ... and not asking leaves it alone
    do64 = res[True][1].double().view(B, T, 8, 64)
    oo = (o.double() + (o_lo.double() if with_lo else 0.0)).view(B, T, 8, 64)
    ref = (do64 * oo).sum(-1).permute(0, 2, 1)                                                         # [B, H, T]
    err = float((res[True][2].double() - ref).abs().max() / (ref.abs().max() + 1e-30))
    assert err < 1e-5, err


def test_tail_backward_delta_refuses_the_lean_form():
    from cosyvoice_lora_finetune_framework_amd.hipops import binding as cb
    import ctypes as C
    t = cb.BlockTailBwdArgs()
    w = torch.zeros(1024, device=DEV)
    for f in ("x1", "dy", "gamma", "mean", "rstd", "z", "W_bwd", "dx1", "dout", "attn_o", "delta"):
        setattr(t, f, cb.ptr(w))
    t.M, t.F, t.DI, t.act, t.lddo, t.ldao, t.T, t.lean = 64, 1024, 512, 3, 512, 512, 32, 1
    assert cb.lib().cvft_block_tail_bwd(C.byref(t), cb.stream()) != 0
