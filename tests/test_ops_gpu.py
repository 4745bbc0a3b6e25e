"""GPU parity tests of every libcvft kernel through the C ABI (ctypes) against plain fp32/fp64
torch references of the same op computed on the CPU.  Tolerances: fp32 path (exact-fp32 MFMA)
relative L2 <= 2e-5; bf16 path <= 2e-2 (bf16 storage)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"
TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-2}
DTYPES = [torch.float32, torch.bfloat16]


def HFmod():
    from cosyvoice_lora_finetune_framework_amd.hipops import functional as HF
    return HF


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def q(t, dtype):
    """round a CPU fp32 tensor to `dtype` and back (reference sees the same stored values)"""
    return t.to(dtype).float()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(37, 80, 80), (300, 512, 256), (1000, 1536, 256), (129, 4097, 192), (5, 16, 320),
                                   (2048, 2048, 512)])
def test_gemm_plain(dtype, M, N, K):
    HF = HFmod()
    x, w, b = q(rnd(M, K, seed=1), dtype), q(rnd(N, K, seed=2) / math.sqrt(K), dtype), rnd(N, seed=3)
    y = HF.gemm(x.to(DEV, dtype), w.to(DEV, dtype), bias=b.to(DEV))
    ref = x.double() @ w.double().t() + b.double()
    assert rel(y, ref) < TOL[dtype]


@pytest.mark.parametrize("M,N,K,r,label", [(4000, 1000, 512, 16, "96,256,3,4"), (5328, 1024, 2048, 0, "96,256,3,4"),
                                           (4100, 1004, 512, 16, "96,256,3,4"), (2100, 1004, 256, 0, "64,64,2,2"),
                                           (4100, 1004, 256, 16, "128,64,4,2"),
                                           (4200, 2052, 512, 48, "128,128,4,2")])
def test_gemm_tile_configs_and_epilogue_forms(M, N, K, r, label):
    """bf16 LDS-DMA GEMM: the one-round 96x256 x 12-wave configuration (ragged DMA piece deal), and both forms of the register
    epilogue under the W image's column map -- 16-byte (N % 8 == 0) and 8-byte (N % 4 == 0 only) -- with bias, activation,
    pre-activation output, residual and a rank extension, partial edge tiles in both dimensions."""
    HF = HFmod()
    dtype = torch.bfloat16
    x, w = q(rnd(M, K, seed=1), dtype), q(rnd(N, K, seed=2) / math.sqrt(K), dtype)
    b, res = rnd(N, seed=5), q(rnd(M, N, seed=6), dtype)
    u = q(rnd(M, r, seed=3), dtype) if r else None
    bl = q(rnd(N, r, seed=4) * 0.1, dtype) if r else None
    pre = torch.empty(M, N, device=DEV, dtype=dtype)
    y = HF.gemm(x.to(DEV, dtype), w.to(DEV, dtype), bias=b.to(DEV), U=None if u is None else u.to(DEV, dtype),
                Bl=None if bl is None else bl.to(DEV, dtype), act="gelu_erf", preact=pre, residual=res.to(DEV, dtype))
    assert label in HF.lib().cvft_gemm_last_kernel().decode(), HF.lib().cvft_gemm_last_kernel().decode()
    z = x.double() @ w.double().t() + b.double()
    if r:
        z = z + u.double() @ bl.double().t()
    assert rel(pre, z) < TOL[dtype]
    assert rel(y, F.gelu(z) + res.double()) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("act", [None, "relu", "silu", "gelu_erf", "gelu_tanh", "mish"])
def test_gemm_lora_epilogue(dtype, act):
    HF = HFmod()
    M, N, K, r = 333, 200, 264, 16
    x, w = q(rnd(M, K, seed=1), dtype), q(rnd(N, K, seed=2) / math.sqrt(K), dtype)
    u, bl, b, res = q(rnd(M, r, seed=3), dtype), q(rnd(N, r, seed=4) * 0.1, dtype), rnd(N, seed=5), q(rnd(M, N, seed=6), dtype)
    pre = torch.empty(M, N, device=DEV, dtype=dtype)
    y = HF.gemm(x.to(DEV, dtype), w.to(DEV, dtype), bias=b.to(DEV), U=u.to(DEV, dtype), Bl=bl.to(DEV, dtype), act=act,
                preact=pre, residual=res.to(DEV, dtype))
    z = x.double() @ w.double().t() + u.double() @ bl.double().t() + b.double()
    acts = {None: lambda t: t, "relu": F.relu, "silu": F.silu, "gelu_erf": F.gelu,
            "gelu_tanh": lambda t: F.gelu(t, approximate="tanh"), "mish": F.mish}
    assert rel(pre, z) < TOL[dtype]
    assert rel(y, acts[act](z) + res.double()) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("act", [None, "gelu_erf", "silu"])
def test_lora_linear_backward(dtype, act):
    """LoRALinear fwd+bwd (reference lora.py:64-76) vs torch autograd."""
    HF = HFmod()
    M, N, K, r, s = 500, 160, 96, 8, 2.0
    x, w, b = q(rnd(M, K, seed=1), dtype), q(rnd(N, K, seed=2) / math.sqrt(K), dtype), rnd(N, seed=3)
    A, Bm = q(rnd(r, K, seed=4) / math.sqrt(K), dtype), q(rnd(N, r, seed=5) * 0.1, dtype)
    res, gy = q(rnd(M, N, seed=6), dtype), q(rnd(M, N, seed=7), dtype)
    xd = x.to(DEV, dtype).requires_grad_(True)
    Ad, Bd = A.to(DEV).requires_grad_(True), Bm.to(DEV).requires_grad_(True)
    rd = res.to(DEV, dtype).requires_grad_(True)
    pack = HF.LinearPack(w.to(DEV), b.to(DEV), dtype)
    y = HF.lora_linear(xd, pack, Ad, Bd, s, act, rd)
    y.backward(gy.to(DEV, dtype))
    xr, Ar, Br, rr = (t.double().requires_grad_(True) for t in (x, A, Bm, res))
    z = xr @ w.double().t() + b.double() + s * (xr @ Ar.t()) @ Br.t()
    a = {None: lambda t: t, "gelu_erf": F.gelu, "silu": F.silu}[act](z)
    (a + rr).backward(gy.double())
    tol = TOL[dtype] * (3 if dtype == torch.bfloat16 else 1)
    assert rel(y, a + rr) < tol
    assert rel(xd.grad, xr.grad) < tol
    assert rel(Ad.grad, Ar.grad) < tol
    assert rel(Bd.grad, Br.grad) < tol
    assert rel(rd.grad, rr.grad) < tol


def _masks(lens, T):
    return (torch.arange(T).unsqueeze(0) < torch.tensor(lens).unsqueeze(1)).float()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kind,T", [("k3s1", 37), ("k3s2", 37), ("k3s2", 40), ("convT", 19), ("k1", 21)])
def test_conv1d_taps(dtype, kind, T):
    """Conv1d k3 s1/s2, k1, ConvTranspose1d(k4,s2,p1) with x*mask pre-multiply: fwd + input grad."""
    HF = HFmod()
    B, Cin, Cout = 3, 40, 24
    lens = [T, T - 5, max(1, T // 2)]
    x = q(rnd(B, Cin, T, seed=1), dtype)
    m = _masks(lens, T).unsqueeze(1)
    if kind == "convT":
        conv = torch.nn.ConvTranspose1d(Cin, Cout, 4, 2, 1)
    elif kind == "k1":
        conv = torch.nn.Conv1d(Cin, Cout, 1)
    else:
        conv = torch.nn.Conv1d(Cin, Cout, 3, 2 if kind == "k3s2" else 1, 1)
    with torch.no_grad():
        conv.weight.copy_(q(conv.weight, dtype))
    xr = x.double().requires_grad_(True)
    conv_d = conv.double()
    yr = conv_d(xr * m.double())
    Tout = yr.shape[-1]
    if kind == "convT":
        Tout = Tout - 1           # exercise the crop-to-skip path
        yr = yr[:, :, :Tout]
    gy = q(rnd(B, Cout, Tout, seed=9), dtype)
    yr.backward(gy.double())
    pack = HF.ConvPack(conv.weight.float().to(DEV), conv.bias.float().to(DEV), dtype, stride=conv.stride[0],
                       transposed=(kind == "convT"))
    xd = x.transpose(1, 2).reshape(B * T, Cin).to(DEV, dtype).requires_grad_(True)
    ln = torch.tensor(lens, dtype=torch.int32, device=DEV)
    y = HF.conv1d(xd, pack, B, T, Tout, in_len=ln)
    y.backward(gy.transpose(1, 2).reshape(B * Tout, Cout).to(DEV, dtype))
    assert rel(y.reshape(B, Tout, Cout).transpose(1, 2), yr) < TOL[dtype]
    assert rel(xd.grad.reshape(B, T, Cin).transpose(1, 2), xr.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("relu,post", [(False, 1.0), (True, 22.627417)])
def test_layernorm(dtype, relu, post):
    HF = HFmod()
    rows, Cn = 77, 256
    x, g, b, gy = q(rnd(rows, Cn, seed=1) * 2 + 0.3, dtype), 1 + 0.1 * rnd(Cn, seed=2), 0.1 * rnd(Cn, seed=3), q(rnd(rows, Cn, seed=4), dtype)
    xd = x.to(DEV, dtype).requires_grad_(True)
    y = HF.layernorm(xd, g.to(DEV), b.to(DEV), 1e-5, relu, post)
    y.backward(gy.to(DEV, dtype))
    xr = x.double().requires_grad_(True)
    yr = F.layer_norm(xr, (Cn,), g.double(), b.double(), 1e-5)
    if relu:
        yr = F.relu(yr)
    yr = yr * post
    yr.backward(gy.double())
    assert rel(y, yr) < TOL[dtype]
    assert rel(xd.grad, xr.grad) < TOL[dtype] * 2


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("G,Cn", [(8, 64), (1, 80), (8, 32)])      # (8, 32): Cg = 4 -> the scalar (non-vector) kernels in bf16
def test_groupnorm_mish(dtype, G, Cn):
    HF = HFmod()
    B, T = 3, 29
    lens = [29, 20, 7]
    x, g, b = q(rnd(B, Cn, T, seed=1) + 0.2, dtype), 1 + 0.1 * rnd(Cn, seed=2), 0.1 * rnd(Cn, seed=3)
    add, gy = q(rnd(B, Cn, seed=4), dtype), q(rnd(B, Cn, T, seed=5), dtype)
    m = _masks(lens, T).unsqueeze(1).double()
    xr = x.double().requires_grad_(True)
    yr = F.mish(F.group_norm(xr, G, g.double(), b.double(), 1e-5)) * m + add.double().unsqueeze(-1)
    yr.backward(gy.double())
    xd = x.transpose(1, 2).reshape(B * T, Cn).to(DEV, dtype).requires_grad_(True)
    y = HF.groupnorm_mish(xd, g.to(DEV), b.to(DEV), B, T, G, 1e-5, torch.tensor(lens, dtype=torch.int32, device=DEV),
                          add.to(DEV, dtype), True)
    y.backward(gy.transpose(1, 2).reshape(B * T, Cn).to(DEV, dtype))
    assert rel(y.reshape(B, T, Cn).transpose(1, 2), yr) < TOL[dtype]
    assert rel(xd.grad.reshape(B, T, Cn).transpose(1, 2), xr.grad) < TOL[dtype] * 3


@pytest.mark.parametrize("dtype", DTYPES)
def test_groupnorm_backward_ignores_bucket_padding_frames(dtype):
    """frames of a shape-bucketed layout beyond the exact batch's frame count (t >= *t_eff) are never normalised
    by the forward, so whatever sits there (here: NaN) must not reach the live frames' gradients -- the backward's group
    reductions are selects, not 0 * x products.  Live-frame dx equals the run on a clean buffer bit for bit; padding dx is 0."""
    HF = HFmod()
    B, T, Te, G, Cn = 2, 40, 29, 8, 256
    lens = torch.tensor([29, 20], dtype=torch.int32, device=DEV)
    te = torch.tensor([Te], dtype=torch.int32, device=DEV)
    g, b = (1 + 0.1 * rnd(Cn, seed=2)).to(DEV), (0.1 * rnd(Cn, seed=3)).to(DEV)
    x0 = q(rnd(B, T, Cn, seed=1) + 0.2, dtype).to(DEV, dtype)
    gy = q(rnd(B, T, Cn, seed=5), dtype).reshape(B * T, Cn).to(DEV, dtype)
    res = []
    for poison in (False, True):
        x = x0.clone()
        if poison:
            x[:, Te:, :] = float("nan")
        xd = x.reshape(B * T, Cn).requires_grad_(True)
        y = HF.groupnorm_mish(xd, g, b, B, T, G, 1e-5, lens, None, True, t_eff=te)
        y.backward(gy)
        res.append((y.detach().reshape(B, T, Cn), xd.grad.reshape(B, T, Cn)))
    assert torch.isfinite(res[1][1][:, :Te]).all()
    assert torch.equal(res[0][0][:, :Te], res[1][0][:, :Te]) and torch.equal(res[0][1][:, :Te], res[1][1][:, :Te])
    assert float(res[1][1][:, Te:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T,lens", [(50, [50, 33]), (131, [131, 64]), (64, [64, 1])])
def test_attn_bias(dtype, T, lens):
    """modules.Attention math (modules.py:253-293) with the additive -1e10 key bias."""
    HF = HFmod()
    B, H = 2, 2
    qkv = q(rnd(B, T, 3 * H * 64, seed=1), dtype)
    gy = q(rnd(B, T, H * 64, seed=2), dtype)
    qr = qkv.double().requires_grad_(True)
    qq, kk, vv = (t.reshape(B, T, H, 64).transpose(1, 2) for t in qr.split(H * 64, dim=-1))
    bias = (1.0 - _masks(lens, T)).double() * -1.0e10
    sim = qq @ kk.transpose(-1, -2) * 0.125 + bias.view(B, 1, 1, T)
    orf = (sim.softmax(-1) @ vv).transpose(1, 2).reshape(B, T, H * 64)
    orf.backward(gy.double())
    qd = qkv.reshape(B * T, -1).to(DEV, dtype).requires_grad_(True)
    o = HF.attn_bias(qd[:, :H * 64], qd[:, H * 64:2 * H * 64], qd[:, 2 * H * 64:], B, H, T,
                     torch.tensor(lens, dtype=torch.int32, device=DEV), 0.125)
    o.backward(gy.reshape(B * T, -1).to(DEV, dtype))
    assert rel(o.reshape(B, T, -1), orf) < TOL[dtype]
    assert rel(qd.grad.reshape(B, T, -1), qr.grad) < TOL[dtype] * 3


@pytest.mark.parametrize("T,lens", [(250, [250, 200, 31, 250]), (500, [500, 420, 77, 500]), (96, [96, 1, 50, 96])])
def test_attn_bias_backward_with_delta_given(T, lens):
    """cvft_attn_bias_bwd with o == NULL: delta = rowsum(dO . (O + O_lo)) per head is an INPUT (the producer of dO formed it: the
    estimator block's tail backward) and neither role reads the forward's output -- against the same call with o / o_lo given
    (which forms delta itself): the gradients agree to the fp32 summation order of delta (both the whole-sequence and the
    per-step staging forms, ragged key lengths)."""
    from cosyvoice_lora_finetune_framework_amd.hipops import binding as cb
    B, H = len(lens), 8
    g = torch.Generator().manual_seed(T)
    q, k, v, do = (torch.randn(B * T, H * 64, generator=g).to(torch.bfloat16).to(DEV) for _ in range(4))
    klen = torch.tensor(lens, dtype=torch.int32, device=DEV)
    o, o_lo = torch.empty_like(q), torch.empty_like(q)
    lse = torch.empty((B, H, T), dtype=torch.float32, device=DEV)
    cb.check(cb.lib().cvft_attn_bias_fwd(cb.dt(q), B, H, T, cb.ptr(q), cb.ptr(k), cb.ptr(v), H * 64, cb.ptr(klen), 0.125, 0, cb.ptr(o), H * 64,
                                         cb.ptr(lse), cb.ptr(o_lo), cb.stream()), "fwd")
    res = []
    for given in (False, True):
        dqkv = torch.zeros((B * T, 3 * H * 64), dtype=torch.bfloat16, device=DEV)
        dq, dk, dv = dqkv[:, :H * 64], dqkv[:, H * 64:2 * H * 64], dqkv[:, 2 * H * 64:]
        if given:
            delta = (do.float() * (o.float() + o_lo.float())).view(B, T, H, 64).sum(-1).permute(0, 2, 1).contiguous()
            po, plo = None, None
        else:
            delta, po, plo = torch.empty((B, H, T), dtype=torch.float32, device=DEV), o, o_lo
        cb.check(cb.lib().cvft_attn_bias_bwd(cb.dt(q), B, H, T, cb.ptr(q), cb.ptr(k), cb.ptr(v), H * 64, cb.ptr(klen), 0.125, 0, cb.ptr(po),
                                             cb.ptr(do), H * 64, cb.ptr(lse), cb.ptr(plo), cb.ptr(delta), cb.ptr(dq), cb.ptr(dk), cb.ptr(dv),
                                             3 * H * 64, cb.stream()), "bwd")
        torch.cuda.synchronize()
        res.append(dqkv)
    assert torch.isfinite(res[1].float()).all()
    assert rel(res[1], res[0]) < 2e-3, rel(res[1], res[0])          # (bf16 outputs: a last-place flip where delta's sum order differs)

@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T,lens,iso", [(50, [50, 33], 7), (131, [131, 64], 64), (131, [131, 40], 65), (64, [64, 1], 1),
                                        (200, [200, 150], 199), (96, [96, 96], 96)])
def test_attn_bias_prompt_isolation(dtype, T, lens, iso):
    """The twin's prompt-isolation bias (modules.py:844-879, 1033-1042): -inf between [0, iso) and [iso, T), added to the
    -1e10 key-padding bias; covers a split inside/at a key tile edge, an utterance that ends before the split, and
    iso >= T (no-op).  Compared on the valid query rows: a padded query row whose visible keys are all padding (utterance
    shorter than the split) is a uniform mean of V in the reference and 0 here -- the estimator multiplies padded frames
    by the mask before anything reads them (decoder.py:240-281) and they receive no gradient, so dO is 0 there."""
    HF = HFmod()
    B, H = 2, 2
    qkv = q(rnd(B, T, 3 * H * 64, seed=1), dtype)
    valid = _masks(lens, T).unsqueeze(-1)
    gy = q(rnd(B, T, H * 64, seed=2), dtype) * valid
    qr = qkv.double().requires_grad_(True)
    qq, kk, vv = (t.reshape(B, T, H, 64).transpose(1, 2) for t in qr.split(H * 64, dim=-1))
    bias = ((1.0 - _masks(lens, T)).double() * -1.0e10).view(B, 1, 1, T).expand(B, 1, T, T).clone()
    if iso < T:
        bias[:, :, iso:, :iso] = float("-inf")
        bias[:, :, :iso, iso:] = float("-inf")
    sim = qq @ kk.transpose(-1, -2) * 0.125 + bias
    orf = (sim.softmax(-1) @ vv).transpose(1, 2).reshape(B, T, H * 64)
    orf.backward(gy.double())
    qd = qkv.reshape(B * T, -1).to(DEV, dtype).requires_grad_(True)
    o = HF.attn_bias(qd[:, :H * 64], qd[:, H * 64:2 * H * 64], qd[:, 2 * H * 64:], B, H, T,
                     torch.tensor(lens, dtype=torch.int32, device=DEV), 0.125, iso if iso < T else 0)
    o.backward(gy.reshape(B * T, -1).to(DEV, dtype))
    assert torch.isfinite(o).all() and torch.isfinite(qd.grad).all()
    assert rel(o.reshape(B, T, -1).cpu() * valid, orf * valid) < TOL[dtype]
    assert rel(qd.grad.reshape(B, T, -1), qr.grad) < TOL[dtype] * 3


@pytest.mark.parametrize("dtype", DTYPES)
def test_masked_mse_frame_weights(dtype):
    """flow_model.py:183-202: loss_mask with the prompt zeroed and a boundary weight; the weight multiplies the residual
    (so it enters squared) and the denominator is sum(loss_mask) * 80.  The boundary band is written past an utterance's
    end too (the reference does not re-mask it)."""
    HF = HFmod()
    B, T, lens, plens, bf, bw = 3, 37, [37, 20, 9], [5, 0, 7], 6, 3.0
    pred = q(rnd(B, T, 80, seed=1), dtype)
    u = rnd(B, T, 80, seed=2)
    w = _masks(lens, T).clone()
    for i, pl in enumerate(plens):
        if pl > 0:
            w[i, :pl] = 0
            w[i, pl:min(pl + bf, T)] = bw
    pr = pred.double().requires_grad_(True)
    lr = (((pr - u.double()) * w.unsqueeze(-1).double()) ** 2).sum() / (w.sum().double() * 80)
    lr.backward()
    pd = pred.reshape(B * T, 80).to(DEV, dtype).requires_grad_(True)
    den = (w.sum() * 80).float().to(DEV)
    loss = HF.masked_mse(pd, u.reshape(B * T, 80).to(DEV), None, den, B, T, weight=w.reshape(-1).float().to(DEV))
    loss.backward()
    assert abs(float(loss) - float(lr)) / float(lr) < (1e-5 if dtype == torch.float32 else 2e-2)
    assert rel(pd.grad.reshape(B, T, 80), pr.grad) < TOL[dtype] * 2


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("L,lens", [(45, [45, 30]), (130, [130, 77]), (64, [64, 64])])
def test_attn_relpos(dtype, causal, L, lens):
    """RelPositionMultiHeadedAttention core (attention.py:276-330, 82-127) vs the oracle's math."""
    from oracle import ref_math as R
    HF = HFmod()
    B, H = 2, 2
    d = H * 64
    qkv = q(rnd(B, L, 3 * d, seed=1), dtype)
    p = q(rnd(2 * L - 1, d, seed=2), dtype)
    bu, bv = 0.3 * rnd(H, 64, seed=3), 0.3 * rnd(H, 64, seed=4)
    gy = q(rnd(B, L, d, seed=5), dtype)
    qr = qkv.double().requires_grad_(True)
    qq, kk, vv = (t.reshape(B, L, H, 64) for t in qr.split(d, dim=-1))
    pr = p.double().requires_grad_(True)
    pp = pr.view(1, -1, H, 64).transpose(1, 2)
    ac = (qq + bu.double()).transpose(1, 2) @ kk.transpose(1, 2).transpose(-1, -2)
    bd = R.rel_shift((qq + bv.double()).transpose(1, 2) @ pp.transpose(-1, -2))
    mask = _masks(lens, L).bool().unsqueeze(1)
    if causal:
        mask = mask & torch.tril(torch.ones(L, L, dtype=torch.bool)).unsqueeze(0)
    mm = mask.unsqueeze(1).eq(0)
    sc = ((ac + bd) / 8.0).masked_fill(mm, -float("inf"))
    at = torch.softmax(sc, -1).masked_fill(mm, 0.0)
    orf = (at @ vv.transpose(1, 2)).transpose(1, 2).reshape(B, L, d)
    orf.backward(gy.double())
    qd = qkv.reshape(B * L, -1).to(DEV, dtype).requires_grad_(True)
    o = HF.attn_relpos(qd[:, :d], qd[:, d:2 * d], qd[:, 2 * d:], p.to(DEV, dtype), bu.to(DEV), bv.to(DEV), B, H, L,
                       torch.tensor(lens, dtype=torch.int32, device=DEV), causal, 0.125)
    o.backward(gy.reshape(B * L, -1).to(DEV, dtype))
    assert rel(o.reshape(B, L, -1), orf) < TOL[dtype]
    assert rel(qd.grad.reshape(B, L, -1), qr.grad) < TOL[dtype] * 3
    # gradient w.r.t. the projected positional encoding (LoRA on linear_pos, lora.py:155-166): same q/k/v gradients, plus dP
    qd2 = qkv.reshape(B * L, -1).to(DEV, dtype).requires_grad_(True)
    pd = p.to(DEV, dtype).requires_grad_(True)
    o2 = HF.attn_relpos(qd2[:, :d], qd2[:, d:2 * d], qd2[:, 2 * d:], pd, bu.to(DEV), bv.to(DEV), B, H, L,
                        torch.tensor(lens, dtype=torch.int32, device=DEV), causal, 0.125)
    o2.backward(gy.reshape(B * L, -1).to(DEV, dtype))
    assert rel(qd2.grad.reshape(B, L, -1), qr.grad) < TOL[dtype] * 3
    assert rel(pd.grad, pr.grad) < TOL[dtype] * 3


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("Lin,Lout", [(13, 24), (290, 500), (9, 17), (24, 24)])
def test_interp_linear(dtype, Lin, Lout):
    HF = HFmod()
    B, Cn = 2, 80
    x, gy = q(rnd(B, Lin, Cn, seed=1), dtype), q(rnd(B, Lout, Cn, seed=2), dtype)
    xr = x.clone().requires_grad_(True)
    yr = F.interpolate(xr.transpose(1, 2), size=Lout, mode="linear").transpose(1, 2)
    yr.backward(gy)
    xd = x.reshape(B * Lin, Cn).to(DEV, dtype).requires_grad_(True)
    y = HF.interp_linear(xd, B, Lin, Lout)
    y.backward(gy.reshape(B * Lout, Cn).to(DEV, dtype))
    assert rel(y.reshape(B, Lout, Cn), yr) < TOL[dtype]
    assert rel(xd.grad.reshape(B, Lin, Cn), xr.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_shape_bucket_scalars_reproduce_exact_shapes(dtype):
    """cvft.h `t_eff` / `eff`: GroupNorm+Mish and the linear interpolation on tensors PADDED to a shape bucket, with the exact
    batch's frame counts passed as device scalars, give bit-for-bit what the exact-shape launch gives on the valid frames
    (forward and input gradient) and zeros on the padding."""
    HF = HFmod()
    B, T, Tp, Cn, G = 3, 29, 36, 64, 8
    lens = torch.tensor([29, 20, 7], dtype=torch.int32, device=DEV)
    x = q(rnd(B, T, Cn, seed=1) + 0.2, dtype).to(DEV, dtype)
    gy = q(rnd(B, T, Cn, seed=5), dtype).to(DEV, dtype)
    g, b, add = (1 + 0.1 * rnd(Cn, seed=2)).to(DEV), (0.1 * rnd(Cn, seed=3)).to(DEV), q(rnd(B, Cn, seed=4), dtype).to(DEV, dtype)
    junk = lambda *sh: (torch.randn(*sh, generator=torch.Generator().manual_seed(9)) * 3).to(DEV, dtype)

    def padT(t, Tn, Tpn):          # [B, Tn, C] -> [B, Tpn, C], padding frames filled with junk (must not matter)
        out = junk(B, Tpn, t.shape[2])
        out[:, :Tn] = t
        return out
    te = torch.tensor([T], dtype=torch.int32, device=DEV)
    for length in (lens, None):
        xe = x.reshape(B * T, Cn).clone().requires_grad_(True)
        ye = HF.groupnorm_mish(xe, g, b, B, T, G, 1e-5, length, add, True)
        ye.backward(gy.reshape(B * T, Cn))
        xp = padT(x, T, Tp).reshape(B * Tp, Cn).requires_grad_(True)
        yp = HF.groupnorm_mish(xp, g, b, B, Tp, G, 1e-5, length, add, True, t_eff=te)
        yp.backward(padT(gy, T, Tp).reshape(B * Tp, Cn))
        yp3, gp3 = yp.reshape(B, Tp, Cn), xp.grad.reshape(B, Tp, Cn)
        assert torch.equal(yp3[:, :T], ye.reshape(B, T, Cn)) and torch.equal(gp3[:, :T], xe.grad.reshape(B, T, Cn))
        assert float(yp3[:, T:].abs().max()) == 0.0 and float(gp3[:, T:].abs().max()) == 0.0
    # interpolation Lin -> Lout, both padded
    Lin, Lout, Lip, Lop = 17, 29, 24, 36
    h = q(rnd(B, Lin, Cn, seed=6), dtype).to(DEV, dtype)
    gz = q(rnd(B, Lout, Cn, seed=7), dtype).to(DEV, dtype)
    he = h.reshape(B * Lin, Cn).clone().requires_grad_(True)
    ze = HF.interp_linear(he, B, Lin, Lout)
    ze.backward(gz.reshape(B * Lout, Cn))
    hp = padT(h, Lin, Lip).reshape(B * Lip, Cn).requires_grad_(True)
    zp = HF.interp_linear(hp, B, Lip, Lop, eff=torch.tensor([Lin, Lout], dtype=torch.int32, device=DEV))
    zp.backward(padT(gz, Lout, Lop).reshape(B * Lop, Cn))
    zp3, gh3 = zp.reshape(B, Lop, Cn), hp.grad.reshape(B, Lip, Cn)
    assert torch.equal(zp3[:, :Lout], ze.reshape(B, Lout, Cn)) and torch.equal(gh3[:, :Lin], he.grad.reshape(B, Lin, Cn))
    assert float(zp3[:, Lout:].abs().max()) == 0.0 and float(gh3[:, Lin:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
def test_cross_entropy(dtype):
    from oracle import ref_math as R
    HF = HFmod()
    n, V = 37, 4097
    lg = q(rnd(n, V, seed=1) * 2, dtype)
    tg = torch.randint(0, V, (n,), generator=torch.Generator().manual_seed(2))
    tg[:5] = -1
    tg[20] = -1
    lr = lg.double().requires_grad_(True)
    loss_r = R.ce_ignore(lr.view(1, n, V), tg.view(1, n))
    loss_r.backward()
    acc_r = R.th_accuracy(lg.view(-1, V), tg.view(1, n))
    ld = lg.to(DEV, dtype).requires_grad_(True)
    loss, acc = HF.cross_entropy(ld, tg.to(DEV, torch.int32))
    (loss * 3.0).backward()
    assert abs(float(loss) - float(loss_r)) / abs(float(loss_r)) < 1e-5
    assert abs(float(acc) - float(acc_r)) < 1e-7
    assert rel(ld.grad, 3.0 * lr.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_resnet_block_fork_gradients_meet_in_the_dgrad_launch(dtype):
    """ResnetBlock1D (matcha decoder.py:76-94): x feeds block1's conv and res_conv.  With HF.CONV_FORK the res_conv backward parks
    its dx and block1's dgrad launch adds it (no autograd accumulation launch); against the same block with the hand-over off
    (autograd's own add) and against torch fp64 autograd on the module's own math; ragged lengths; nothing left parked."""
    HF = HFmod()
    from cosyvoice_lora_finetune_framework_amd.modules import ResnetBlock1D
    torch.manual_seed(4)
    B, T, Cin, Cout, Ct = 3, 50, 320, 256, 64
    blk = ResnetBlock1D(Cin, Cout, Ct).to(DEV)
    length = torch.tensor([50, 37, 12], dtype=torch.int32, device=DEV)
    x0 = torch.randn(B * T, Cin, device=DEV).to(dtype)
    temb = torch.randn(B, Ct, device=DEV).to(dtype)
    g = torch.randn(B * T, Cout, device=DEV).to(dtype)

    def run(fork: bool):
        HF.CONV_FORK = fork
        HF._PRE_DX.clear()
        taken = HF.HANDS_TAKEN.get(HF._H_FORK, 0)
        x = x0.clone().requires_grad_(True)
        y = blk(x, B, T, length, torch.nn.functional.mish(temb.float()).to(dtype))
        y.backward(g)
        torch.cuda.synchronize()
        # the pairing rode on x (taken by the "park" conv's forward) and the parked gradient was taken under its token
        assert HF.HANDS_TAKEN.get(HF._H_FORK, 0) - taken == (1 if fork else 0)
        assert HF._H_FORK not in x.__dict__ and len(HF._PRE_DX) == 0
        return y.detach().clone(), x.grad.clone()
    default = HF.CONV_FORK
    try:
        ya, ga = run(True)
        yb, gb = run(False)
    finally:
        HF.CONV_FORK = default
    assert torch.equal(ya, yb)
    assert rel(ga, gb) < (1e-6 if dtype == torch.float32 else 4e-3)
    # fp64 torch on the same weights: mask -> conv -> GroupNorm -> Mish -> mask (+ time term), twice, plus the 1x1 residual conv
    import torch.nn.functional as F
    xr = x0.double().reshape(B, T, Cin).transpose(1, 2).clone().requires_grad_(True)
    mask = (torch.arange(T, device=DEV)[None, :] < length[:, None]).double()[:, None, :]
    W = lambda m: (m.weight.double(), m.bias.double())

    def block1d(b1d, h):
        conv, gn = b1d.block[0], b1d.block[1]
        h = F.conv1d(h * mask, *W(conv), padding=1)
        h = F.group_norm(h, gn.num_groups, gn.weight.double(), gn.bias.double(), gn.eps)
        return F.mish(h) * mask
    h = block1d(blk.block1, xr)
    h = h + F.linear(F.mish(temb.double()), *W(blk.mlp[1]))[:, :, None]
    h = block1d(blk.block2, h)
    yr = h + F.conv1d(xr * mask, *W(blk.res_conv))
    yr.backward(g.double().reshape(B, T, Cout).transpose(1, 2))
    gr = xr.grad.transpose(1, 2).reshape(B * T, Cin)
    assert rel(ga, gr) < (2e-5 if dtype == torch.float32 else 3e-2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("V,pitch", [(4097, 4160), (4097, 4104), (515, 520), (9000, 9008)])
def test_cross_entropy_padded_pitch_rows(dtype, V, pitch):
    """The LLM's logits arrive as a [n, V] view of a zero-padded [n, pitch] buffer (16-byte aligned rows): cvft_ce_fwd takes a wave
    per row with 16-byte loads -- the row held in registers up to 64 * 16 chunks, the two-pass loop beyond (V = 9000 in bf16 stays in
    registers, in fp32 it does not) -- and more rows than 4 x 512 waves so that waves walk several rows; cvft_ce_bwd writes 16-byte
    chunks.  Against torch in fp64 on the same (rounded) logits: loss, accuracy with torch.argmax's first-index tie rule, gradient;
    the pad columns of the gradient stay zero."""
    HF = HFmod()
    n = 2100 if V < 5000 else 300
    g = torch.Generator().manual_seed(3)
    lg = q(torch.randn(n, V, generator=g) * 2, dtype)
    tg = torch.randint(0, V, (n,), generator=g)
    tg[::7] = -1
    # ties: the maximum twice in a row; the target is the first of the two (counts as correct) or the second (does not)
    for r, (i, j, t) in {1: (10, 200, 10), 2: (10, 200, 200), 5: (V - 9, V - 1, V - 9), 6: (V - 9, V - 1, V - 1)}.items():
        lg[r, i] = lg[r, j] = 30.0
        tg[r] = t
    base = torch.zeros(n, pitch, device=DEV, dtype=dtype)
    base[:, :V] = lg.to(DEV, dtype)
    ld = base[:, :V].detach().requires_grad_(True)
    assert ld.stride(0) == pitch
    loss, acc = HF.cross_entropy(ld, tg.to(DEV, torch.int32), 0.0)
    (loss * 2.0).backward()
    lr = lg.double().requires_grad_(True)
    valid = tg >= 0
    loss_r = torch.nn.functional.cross_entropy(lr[valid], tg[valid])
    (loss_r * 2.0).backward()
    acc_r = (lr.detach().argmax(1)[valid] == tg[valid]).double().mean()
    assert abs(float(loss) - float(loss_r)) / abs(float(loss_r)) < 2e-5
    assert abs(float(acc) - float(acc_r)) < 1e-6
    assert rel(ld.grad, lr.grad) < TOL[dtype]
    assert float(ld.grad[~valid.to(DEV)].abs().max()) == 0.0
    gbase = ld.grad.as_strided((n, pitch), (ld.grad.stride(0), 1)) if ld.grad.stride(0) == pitch else None
    if gbase is not None:
        assert float(gbase[:, V:].abs().max()) == 0.0


def test_cross_entropy_label_smoothing_matches_reference():
    """cvft_ce_fwd/bwd with smoothing > 0 against the REFERENCE's LabelSmoothingLoss outputs (tests/golden/lsm.npz: loss and
    autograd gradient, fp32 1e-5) and, at the LLM's vocabulary size in both dtypes, against the oracle restatement."""
    from oracle import ref_math as R
    from conftest import load_npz
    HF = HFmod()
    g = load_npz("lsm.npz")
    n, V = g["logits"].shape[0] * g["logits"].shape[1], g["logits"].shape[2]
    for eps in (0.1, 0.3, 1.0):
        ld = g["logits"].reshape(n, V).to(DEV).requires_grad_(True)
        loss, _ = HF.cross_entropy(ld, g["target"].reshape(n).to(DEV, torch.int32), eps)
        loss.backward()
        assert abs(float(loss) - float(g[f"loss_eps{eps}_tok"])) / abs(float(g[f"loss_eps{eps}_tok"])) < 1e-5
        assert rel(ld.grad.reshape(g["logits"].shape), g[f"grad_eps{eps}_tok"]) < 1e-5
    n, V = 37, 4097
    tg = torch.randint(0, V, (n,), generator=torch.Generator().manual_seed(2))
    tg[:5] = -1
    for dtype in DTYPES:
        lg = q(rnd(n, V, seed=1) * 2, dtype)
        lr = lg.double().requires_grad_(True)
        loss_r = R.ce_label_smoothing(lr.view(1, n, V), tg.view(1, n), 0.1)
        loss_r.backward()
        ld = lg.to(DEV, dtype).requires_grad_(True)
        loss, _ = HF.cross_entropy(ld, tg.to(DEV, torch.int32), 0.1)
        (loss * 3.0).backward()
        assert abs(float(loss) - float(loss_r)) / abs(float(loss_r)) < (1e-5 if dtype == torch.float32 else 1e-4)
        assert rel(ld.grad, 3.0 * lr.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_cfm_prepare_and_mse(dtype):
    from oracle import ref_math as R
    HF = HFmod()
    B, T = 3, 21
    lens = [21, 15, 9]
    feat = rnd(B, T, 80, seed=1) * 2 - 6
    z = rnd(B, 80, T, seed=2)
    t_raw = torch.rand(B, 1, 1, generator=torch.Generator().manual_seed(3))
    keep = torch.tensor([1.0, 0.0, 1.0])
    mu, spk = q(rnd(B, T, 80, seed=4), dtype), q(rnd(B, 80, seed=5), dtype)
    x1 = ((feat + 6.0) / 2.0).transpose(1, 2)
    t, y, u = R.cfm_prepare(x1, t_raw, z, 1e-6)
    mud = mu.reshape(B * T, 80).to(DEV, dtype).requires_grad_(True)
    xin, ud, td = HF.cfm_prepare(mud, spk.to(DEV, dtype), feat.to(DEV), z.transpose(1, 2).contiguous().to(DEV),
                                 t_raw.view(B).to(DEV), keep.to(DEV), B, T, -6.0, 2.0, 1e-6)
    xin_r = torch.cat([y.transpose(1, 2), mu * keep.view(B, 1, 1), (spk * keep.view(B, 1)).unsqueeze(1).expand(B, T, 80),
                       torch.zeros(B, T, 80)], dim=-1)
    assert rel(xin.reshape(B, T, 320), xin_r) < TOL[dtype]
    assert rel(ud.reshape(B, T, 80), u.transpose(1, 2)) < 1e-6
    assert rel(td, t.view(B)) < 1e-6
    # t_scheduler other than 'cosine' (flow_matching.py:176): t stays as drawn
    tl, yl, _ = R.cfm_prepare(x1, t_raw, z, 1e-6, t_scheduler='linear')
    xl, _, tdl = HF.cfm_prepare(mud.detach(), spk.to(DEV, dtype), feat.to(DEV), z.transpose(1, 2).contiguous().to(DEV),
                                t_raw.view(B).to(DEV), keep.to(DEV), B, T, -6.0, 2.0, 1e-6, None, False)
    assert torch.equal(tdl.cpu(), t_raw.view(B)) and rel(xl.reshape(B, T, 320)[..., :80], yl.transpose(1, 2)) < TOL[dtype]
    # masked MSE through the packed input (grad reaches mu through the keep mask)
    pred = xin[:, 80:160] * 1.5
    ln = torch.tensor(lens, dtype=torch.int32, device=DEV)
    den = (ln.sum() * 80).float()
    loss = HF.masked_mse(pred, ud, ln, den, B, T)
    loss.backward()
    mr = mu.double().requires_grad_(True)
    m = _masks(lens, T).unsqueeze(-1).double()
    pr = mr * keep.view(B, 1, 1).double() * 1.5
    lr = (((pr - u.transpose(1, 2).double()) * m) ** 2).sum() / (m.sum() * 80)
    lr.backward()
    assert abs(float(loss) - float(lr)) / float(lr) < (1e-5 if dtype == torch.float32 else 2e-2)
    assert rel(mud.grad.reshape(B, T, 80), mr.grad) < TOL[dtype] * 2


@pytest.mark.parametrize("dtype", DTYPES)
def test_small_ops(dtype):
    from oracle import ref_math as R
    HF = HFmod()
    # embedding gather with clamp + length mask
    table = q(rnd(50, 32, seed=1), dtype)
    tok = torch.tensor([[3, 7, -1, 49], [0, 5, 9, 2]])
    ln = torch.tensor([4, 2], dtype=torch.int32)
    out = HF.embed_gather(tok.to(DEV), table.to(DEV, dtype), ln.to(DEV))
    ref = F.embedding(tok.clamp(min=0), table) * _masks([4, 2], 4).unsqueeze(-1)
    assert rel(out.reshape(2, 4, 32), ref) < 1e-7
    # ragged row gather / scatter
    src = q(rnd(10, 16, seed=2), dtype)
    idx = torch.tensor([4, -1, 0, 9, -1, 2], dtype=torch.int32)
    sd = src.to(DEV, dtype).requires_grad_(True)
    g = HF.gather_rows(sd, idx.to(DEV), -1.0)
    refg = torch.where(idx.view(-1, 1) >= 0, src[idx.clamp(min=0).long()], torch.tensor(-1.0))
    assert rel(g, refg) < 1e-7
    g.backward(torch.ones_like(g))
    gr = torch.zeros(10, 16)
    gr[idx[idx >= 0].long()] = 1.0
    assert rel(sd.grad, gr) < 1e-7
    # l2 normalise
    e = rnd(5, 192, seed=3)
    assert rel(HF.l2norm_rows(e.to(DEV), dtype), F.normalize(e, dim=1)) < TOL[dtype]
    # sinusoidal time embedding, scale 1000
    from cosyvoice_lora_finetune_framework_amd.modules import SinusoidalPosEmb
    t = torch.tensor([0.0, 0.137, 0.5, 0.999])
    emb = SinusoidalPosEmb(320)(t.to(DEV), dtype=torch.float32)
    assert float((emb.cpu() - R.sinusoidal_pos_emb(t, 320)).abs().max()) < 2e-4


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("causal", [False, True])
def test_dwconv1d(dtype, causal):
    """depthwise Conv1d k=15 (convolution.py:62-70,118), LDS-staged frame tiles."""
    HF = HFmod()
    B, T, Cn, Kw = 2, 70, 96, 15
    x, w, b = q(rnd(B, Cn, T, seed=1), dtype), rnd(Cn, 1, Kw, seed=2) * 0.3, rnd(Cn, seed=3) * 0.1
    gy = q(rnd(B, Cn, T, seed=4), dtype)
    xr = x.double().requires_grad_(True)
    xp = F.pad(xr, (Kw - 1, 0)) if causal else xr
    yr = F.conv1d(xp, w.double(), b.double(), padding=0 if causal else (Kw - 1) // 2, groups=Cn)
    yr.backward(gy.double())
    xd = x.transpose(1, 2).reshape(B * T, Cn).to(DEV, dtype).requires_grad_(True)
    y = HF.dwconv1d(xd, w.reshape(Cn, Kw).to(DEV), b.to(DEV), B, T, Kw - 1 if causal else (Kw - 1) // 2)
    y.backward(gy.transpose(1, 2).reshape(B * T, Cn).to(DEV, dtype))
    assert rel(y.reshape(B, T, Cn).transpose(1, 2), yr) < TOL[dtype]
    assert rel(xd.grad.reshape(B, T, Cn).transpose(1, 2), xr.grad) < TOL[dtype]


def test_adamw_flat_matches_torch():
    from cosyvoice_lora_finetune_framework_amd.hipops import binding as cb
    n = 10007
    p0, g0 = rnd(n, seed=1), rnd(n, seed=2) * 3
    pr = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pr], lr=1e-3, weight_decay=0.01, betas=(0.9, 0.999))
    p, m, v = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    lr = torch.tensor([1e-3], device=DEV)
    for step in range(1, 4):
        g = g0 * step
        pr.grad = g.clone()
        gn = torch.nn.utils.clip_grad_norm_([pr], 1.0)
        opt.step()
        gd = g.to(DEV)
        ss = torch.zeros(1, device=DEV)
        cb.check(cb.lib().cvft_sumsq(n, cb.ptr(gd), cb.ptr(ss), cb.stream()))
        assert abs(math.sqrt(float(ss)) - float(g.norm())) / float(g.norm()) < 1e-5
        st = torch.tensor([float(step)], device=DEV)
        cb.check(cb.lib().cvft_adamw_flat(n, cb.ptr(p), cb.ptr(gd), cb.ptr(m), cb.ptr(v), cb.ptr(lr), 0.9, 0.999, 1e-8,
                                          0.01, cb.ptr(st), cb.ptr(ss), 1.0, 1.0, cb.stream()))
        assert rel(p, pr.data) < 1e-6


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("r", [4, 16, 64, 6])
def test_lora_rank_accum(dtype, r):
    """dA = V^T X and dB = dY^T U on the VALU rank kernel (r % 4 == 0) / MFMA fallback (r = 6)."""
    HF = HFmod()
    M, Cn = 1237, 200
    wd, rk = q(rnd(M, Cn, seed=1), dtype), q(rnd(M, r, seed=2), dtype)
    ref = rk.double().t() @ wd.double()
    out = torch.ones(r, Cn, device=DEV)
    HF.rank_accum(wd.to(DEV, dtype), rk.to(DEV, dtype), out, False)
    assert rel(out - 1.0, ref) < 1e-5
    out_t = torch.zeros(Cn, r, device=DEV)
    HF.rank_accum(wd.to(DEV, dtype), rk.to(DEV, dtype), out_t, True)
    assert rel(out_t, ref.t()) < 1e-5


def test_flat_adamw_shadows_and_direct_grads():
    """FlatAdamW: flat views, bf16 + transposed shadows refreshed per step, LoRA grads accumulated in place."""
    from cosyvoice_lora_finetune_framework_amd.optim import FlatAdamW
    HF = HFmod()
    lin = torch.nn.Linear(96, 160).to(DEV)
    A = torch.nn.Parameter(rnd(8, 96, seed=1).to(DEV) * 0.1)
    Bm = torch.nn.Parameter(rnd(160, 8, seed=2).to(DEV) * 0.1)
    opt = FlatAdamW([A, Bm], lr=1e-2)
    assert A.data_ptr() == opt.flat_p.data_ptr() and A.grad.data_ptr() == opt.flat_g.data_ptr()
    assert rel(A._cvft_shadow[0], A) < 4e-3 and rel(A._cvft_shadow[1], A.t()) < 4e-3
    assert rel(Bm._cvft_shadow[1], Bm.t()) < 4e-3
    pack = HF.LinearPack(lin.weight, lin.bias, torch.bfloat16)
    x = rnd(300, 96, seed=3).to(DEV, torch.bfloat16).requires_grad_(True)
    for it in range(2):
        y = HF.lora_linear(x, pack, A, Bm, 2.0)
        y.float().pow(2).mean().backward()
        g0 = opt.flat_g.clone()
        assert float(g0.abs().sum()) > 0
        a_before = A.detach().clone()
        opt.step()
        opt.zero_grad()
        assert float((A.detach() - a_before).abs().sum()) > 0
        assert rel(A._cvft_shadow[0], A) < 4e-3 and rel(Bm._cvft_shadow[1], Bm.t()) < 4e-3
        assert float(opt.flat_g.abs().sum()) == 0.0


def test_lora_grad_sink_matches_atomic_path_and_is_deterministic():
    """two-stage slabs + single reduce launch == atomic kernel (to fp32 rounding), bitwise reproducible."""
    from cosyvoice_lora_finetune_framework_amd.optim import FlatAdamW
    HF = HFmod()
    lin = torch.nn.Linear(256, 512).to(DEV)
    A = torch.nn.Parameter(rnd(16, 256, seed=1).to(DEV) * 0.1)
    Bm = torch.nn.Parameter(rnd(512, 16, seed=2).to(DEV) * 0.1)
    opt = FlatAdamW([A, Bm], lr=1e-2)
    pack = HF.LinearPack(lin.weight, lin.bias, torch.bfloat16)
    x = rnd(1500, 256, seed=3).to(DEV, torch.bfloat16).requires_grad_(True)
    gy = rnd(1500, 512, seed=4).to(DEV, torch.bfloat16)
    HF.lora_linear(x, pack, A, Bm, 2.0).backward(gy)
    g_atomic = opt.flat_g.clone()
    runs = []
    for _ in range(2):
        opt.zero_grad()
        with HF.LoraGradSink():
            HF.lora_linear(x, pack, A, Bm, 2.0).backward(gy)
        runs.append(opt.flat_g.clone())
    assert torch.equal(runs[0], runs[1])
    assert rel(runs[0], g_atomic) < 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,r", [(333, 200, 264, 16), (4000, 512, 256, 16), (700, 1024, 4096, 16), (129, 96, 64, 8)])
def test_gemm_fused_lora_side_path(dtype, M, N, K, r):
    """x W^T + b + (s x A^T) B^T with U computed INSIDE the launch (cvft_gemm La/Uout) == two-launch form."""
    HF = HFmod()
    x, w = q(rnd(M, K, seed=1), dtype), q(rnd(N, K, seed=2) / math.sqrt(K), dtype)
    A, Bm, b = q(rnd(r, K, seed=3) / math.sqrt(K), dtype), q(rnd(N, r, seed=4) * 0.1, dtype), rnd(N, seed=5)
    U = torch.full((M, r), 7.0, device=DEV, dtype=dtype)
    y = HF.gemm(x.to(DEV, dtype), w.to(DEV, dtype), bias=b.to(DEV), La=A.to(DEV, dtype), lora_scale=2.0, Uout=U,
                Bl=Bm.to(DEV, dtype), act="silu")
    ur = 2.0 * (x.double() @ A.double().t())
    assert rel(U, ur) < TOL[dtype]
    uq = q(ur.float(), dtype).double()            # the kernel feeds the rounded U to the extension step
    ref = F.silu(x.double() @ w.double().t() + b.double() + uq @ Bm.double().t())
    assert rel(y, ref) < TOL[dtype]


@pytest.mark.parametrize("fuse_side", [False, True])
def test_qkv_stacked_matches_separate_projections(fuse_side):
    """bf16 training path: q|k|v as one stacked projection (block-diagonal LoRA-B, rank 48 slab kernels, sub-block
    reduce tasks) must give the same outputs, input gradient and six adapter gradients as three LoRALinear modules."""
    from cosyvoice_lora_finetune_framework_amd.lora import LoRALinear
    from cosyvoice_lora_finetune_framework_amd.modules import hip_qkv
    from cosyvoice_lora_finetune_framework_amd.optim import FlatAdamW
    HF = HFmod()
    torch.manual_seed(3)
    K, N, M = 256, 512, 1000
    mods = [LoRALinear(torch.nn.Linear(K, N), r=16, lora_alpha=32, lora_dropout=0.0).to(DEV) for _ in range(3)]
    for m in mods:
        torch.nn.init.normal_(m.lora_B, std=0.05)
        m.eval()
    params = [p for m in mods for p in (m.lora_A, m.lora_B)]
    opt = FlatAdamW(params, lr=1e-3)
    x = (torch.randn(M, K, device=DEV) * 0.5).to(torch.bfloat16).requires_grad_(True)
    gq, gk, gv = (torch.randn(M, N, device=DEV).to(torch.bfloat16) for _ in range(3))

    def run(stack: bool):
        HF.QKV_STACKING = stack
        HF.QKV_FUSE_SIDE = fuse_side
        opt.zero_grad()
        x.grad = None
        with HF.LoraGradSink():
            q, k, v = hip_qkv(mods[0], mods[1], mods[2], x)
            fused = torch.cat([gq, gk, gv], 1)          # like the attention backward's dq|dk|dv buffer
            torch.autograd.backward([q, k, v], [fused[:, :N], fused[:, N:2 * N], fused[:, 2 * N:]])
        return [t.float().clone() for t in (q, k, v, x.grad)] + [p.grad.clone() for p in params]
    try:
        a = run(True)
        b = run(False)
    finally:
        HF.QKV_STACKING = True
        HF.QKV_FUSE_SIDE = False
    names = ["q", "k", "v", "dx"] + [f"g{i}" for i in range(6)]
    for n, u, w in zip(names, a, b):
        assert rel(u, w) < 2e-2, (n, rel(u, w))
    # and against plain fp32 torch math
    for i, m in enumerate(mods):
        W, bb = m.original_layer.weight.float(), m.original_layer.bias.float()
        ref = x.detach().float() @ W.t() + bb + m.scaling * (x.detach().float() @ m.lora_A.t()) @ m.lora_B.t()
        assert rel(a[i], ref) < 2e-2


@pytest.mark.parametrize("dtype", DTYPES)
def test_dropout_add_statistics_and_backward_mask(dtype):
    """cvft_dropout_add: keep rate ~ 1-p, kept values scaled by 1/(1-p), residual passes through, backward re-derives the
    SAME mask from (seed, site); a new step (seed + 1) draws a different mask."""
    HF = HFmod()
    n, p = 1 << 18, 0.1
    HF.dropout_begin_step()
    x = torch.ones(n, device=DEV, dtype=dtype, requires_grad=True)
    res = torch.full((n,), 3.0, device=DEV, dtype=dtype, requires_grad=True)
    y = HF.dropout_add(x, p, res)
    kept = (y.float() - 3.0)
    rate = float((kept > 0).float().mean())
    assert abs(rate - (1 - p)) < 5e-3, rate
    assert torch.allclose(kept[kept > 0], torch.full_like(kept[kept > 0], 1 / (1 - p)), rtol=1e-2 if dtype == torch.float32 else 3e-2)
    y.sum().backward()
    assert torch.equal((x.grad.float() > 0), (kept > 0))                 # same mask in backward
    assert torch.allclose(res.grad.float(), torch.ones_like(res.grad.float()))
    y2 = HF.dropout_add(x.detach(), p)                                   # next call site, same step: different mask
    assert not torch.equal(y2 > 0, kept > 0)
    HF.dropout_begin_step()
    assert HF.dropout_add(x.detach(), 0.0) is not None


@pytest.mark.parametrize("dtype,M,N,K,r", [(torch.float32, 300, 128, 48, 0), (torch.bfloat16, 2048, 1024, 1024, 16),
                                         (torch.bfloat16, 5328, 1024, 4096, 16), (torch.bfloat16, 500, 260, 64, 0),
                                         (torch.bfloat16, 640, 512, 2048, 16), (torch.float32, 77, 36, 20, 4)])
def test_linear_output_dropout_in_gemm_epilogue(dtype, M, N, K, r):
    """x = residual + dropout(linear_out(.)) (encoder_layer.py:95 / 104) with the mask applied in the GEMM epilogue
    (cvft_gemm odrop_p / odrop_site) against the two-launch form (GEMM, then cvft_dropout_add) under the SAME seed and
    site: identical keep pattern, values and all gradients equal to rounding; every epilogue form (LDS store, 8-byte and
    16-byte register epilogue) is hit by one of the shapes."""
    from cosyvoice_lora_finetune_framework_amd.lora import LoRALinear
    from cosyvoice_lora_finetune_framework_amd.modules import hip_linear
    HF = HFmod()
    torch.manual_seed(5)
    lin = torch.nn.Linear(K, N)
    mod = (LoRALinear(lin, r=r, lora_alpha=2 * r, lora_dropout=0.0) if r else lin).to(DEV)
    if r:
        torch.nn.init.normal_(mod.lora_B, std=0.05)
    mod.train()
    x = (torch.randn(M, K, device=DEV) * 0.5).to(dtype).requires_grad_(True)
    res = torch.randn(M, N, device=DEV).to(dtype).requires_grad_(True)
    g = torch.randn(M, N, device=DEV).to(dtype)
    params = [q for q in mod.parameters() if q.requires_grad]
    HF.dropout_begin_step()
    p = 0.1

    def run(fuse: bool):
        HF.OUT_DROP_FUSE = fuse
        HF._DROPOUT["site"] = 7
        for t in [x, res] + params:
            t.grad = None
        y = hip_linear(mod, x, residual=res, out_drop=p)
        y.backward(g)
        return [y.detach().float(), x.grad.float().clone(), res.grad.float().clone()] + [q.grad.float().clone() for q in params if q.grad is not None]
    try:
        a = run(True)
        b = run(False)
    finally:
        HF.OUT_DROP_FUSE = True
    plain = hip_linear(mod.eval(), x.detach()).float()
    kept_a = (a[0] - res.detach().float()).abs() > 1e-3 * plain.abs().clamp_min(1e-2)
    kept_b = (b[0] - res.detach().float()).abs() > 1e-3 * plain.abs().clamp_min(1e-2)
    big = plain.abs() > 0.05                       # (tiny outputs can round to the residual in bf16)
    assert torch.equal(kept_a[big], kept_b[big])
    rate = float(kept_a[big].float().mean())
    assert abs(rate - (1 - p)) < 2e-2, rate
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for i, (u, w) in enumerate(zip(a, b)):
        assert rel(u, w) < tol, (i, rel(u, w))
    ref = res.detach().float() + torch.where(kept_b, plain / (1 - p), torch.zeros_like(plain))
    assert rel(a[0], ref) < (1e-4 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("act", ["relu", "silu"])
@pytest.mark.parametrize("lora,lora_p", [(True, 0.15), (True, 0.0), (False, 0.0)])
def test_feed_forward_train_mode_fused_matches_unfused(act, lora, lora_p):
    """PositionwiseFeedForward in train mode (positionwise_feed_forward.py:54, encoder_layer.py:104): the one-Function
    form (activation + inner mask in W1's epilogue, outer mask + residual in W2's, act' + inner mask in the epilogue of W2's
    dgrad -- with the masked rank extension of W2's lora_dropout in the same launch) against the launch-per-stage form
    (linear, cvft_act_dropout, linear + epilogue mask) under the SAME seed and mask sites: outputs and every gradient agree
    to bf16 rounding."""
    from cosyvoice_lora_finetune_framework_amd.lora import LoRALinear
    from cosyvoice_lora_finetune_framework_amd.modules import PositionwiseFeedForward
    from cosyvoice_lora_finetune_framework_amd.optim import FlatAdamW
    HF = HFmod()
    torch.manual_seed(21)
    d, hid, M = 256, 1024, 1500
    ff = PositionwiseFeedForward(d, hid, 0.1, act)
    if lora:
        ff.w_1 = LoRALinear(ff.w_1, r=16, lora_alpha=32, lora_dropout=lora_p)
        ff.w_2 = LoRALinear(ff.w_2, r=16, lora_alpha=32, lora_dropout=lora_p)
        torch.nn.init.normal_(ff.w_1.lora_B, std=0.05)
        torch.nn.init.normal_(ff.w_2.lora_B, std=0.05)
    ff = ff.to(DEV).train()
    params = [q for q in ff.parameters() if q.requires_grad] if lora else []
    opt = FlatAdamW(params, lr=1e-3) if lora else None
    x = (torch.randn(M, d, device=DEV) * 0.7).to(torch.bfloat16).requires_grad_(True)
    res = torch.randn(M, d, device=DEV).to(torch.bfloat16).requires_grad_(True)
    g = torch.randn(M, d, device=DEV).to(torch.bfloat16)
    HF.dropout_begin_step()

    def run(fused: bool):
        HF.FFN_TRAIN_FUSE = fused
        HF._DROPOUT["site"] = 40
        x.grad = res.grad = None
        if opt is not None:
            opt.zero_grad()
        with HF.LoraGradSink():
            y = ff(x, res, out_dropout=0.1)
            assert ("FeedForward" in type(y.grad_fn).__name__) == fused
            y.backward(g)
        torch.cuda.synchronize()
        return [y.detach().float(), x.grad.float().clone(), res.grad.float().clone()] + [q.grad.float().clone() for q in params]
    try:
        a = run(True)
        b = run(False)
    finally:
        HF.FFN_TRAIN_FUSE = True
    for i, (u, w) in enumerate(zip(a, b)):
        assert rel(u, w) < 2e-2, (i, rel(u, w))
    # and the masks really are on: eval mode differs
    ff.eval()
    assert rel(ff(x.detach(), res.detach()), a[0]) > 5e-2


@pytest.mark.parametrize("dtype", DTYPES)
def test_layernorm_backward_writes_the_producer_mask_copy(dtype):
    """x = residual + dropout(linear(h)) -> (x, LN(x)) (encoder_layer.py:95-106): the LayerNormForkFn backward also writes
    keep / (1 - p) * dx for the linear in front of it (cvft_layernorm_bwd_mask), which then skips its own mask pass over dx.
    Against the same chain with that hand-over switched off, same seed and sites: every gradient bit-identical; and the
    hand-over really happened (no entry left parked, one cvft_dropout_add launch fewer is visible as an empty registry)."""
    HF = HFmod()
    torch.manual_seed(5)
    M, d, hid = 333, 256, 512
    lin = torch.nn.Linear(hid, d).to(DEV)
    pack = HF.LinearPack(lin.weight, lin.bias, dtype)
    gamma, beta = (torch.rand(d, device=DEV) + 0.5), torch.randn(d, device=DEV) * 0.1
    h = torch.randn(M, hid, device=DEV).to(dtype).requires_grad_(True)
    res = torch.randn(M, d, device=DEV).to(dtype).requires_grad_(True)
    g1, g2 = torch.randn(M, d, device=DEV).to(dtype), torch.randn(M, d, device=DEV).to(dtype)
    HF.dropout_begin_step()

    def run(handover: bool):
        HF.LN_BWD_MASK = handover
        HF._DROPOUT["site"] = 7
        t0 = dict(HF.HANDS_TAKEN)
        h.grad = res.grad = None
        x = HF.lora_linear(h, pack, residual=res, out_drop_p=0.1)
        assert (HF._H_ODROP in x.__dict__) == handover       # the producer's (p, site) rides on its output ...
        xr, xn = HF.layernorm_fork(x, gamma, beta, 1e-5)
        assert HF._H_ODROP not in x.__dict__                 # ... and was taken by the LayerNorm (or never noted)
        torch.autograd.backward([xr, xn], [g1, g2])
        torch.cuda.synchronize()
        took = {k: HF.HANDS_TAKEN.get(k, 0) - t0.get(k, 0) for k in (HF._H_ODROP, HF._H_PRE_MASKED)}
        assert took == ({HF._H_ODROP: 1, HF._H_PRE_MASKED: 1} if handover else {HF._H_ODROP: 0, HF._H_PRE_MASKED: 0}), took
        return x.detach().clone(), h.grad.clone(), res.grad.clone()
    try:
        a = run(True)
        b = run(False)
    finally:
        HF.LN_BWD_MASK = True
    for u, w in zip(a, b):
        assert torch.equal(u, w)
    # the mask is on: about p of the rows' branch gradient is zero -> compare with p = 0
    HF._DROPOUT["site"] = 7
    h.grad = None
    x0 = HF.lora_linear(h, pack, residual=res)
    xr, xn = HF.layernorm_fork(x0, gamma, beta, 1e-5)
    torch.autograd.backward([xr, xn], [g1, g2])
    assert rel(h.grad, a[1]) > 5e-2


@pytest.mark.parametrize("lora_p", [0.0, 0.1])
def test_layernorm_backward_writes_the_adapter_side_product(lora_p):
    """The linear in front of the LayerNorm carries a rank-16 adapter (lora.py:64-76): the LayerNorm backward that writes its
    masked incoming gradient dxm also forms V = s * dxm B (cvft_layernorm_bwd_mask_side) and the adapter's backward takes it
    instead of launching the product.  Checked against fp32 torch on the same dxm (bf16 output: 2^-8 relative per entry), and the
    whole chain against the same chain with the hand-over off (same masks; V differs by summation order only)."""
    HF = HFmod()
    dtype = torch.bfloat16
    torch.manual_seed(11)
    M, d, hid, r = 333, 1024, 512, 16
    lin = torch.nn.Linear(hid, d).to(DEV)
    pack = HF.LinearPack(lin.weight, lin.bias, dtype)
    A = (torch.randn(r, hid, device=DEV) * 0.05).requires_grad_(True)
    Bm = (torch.randn(d, r, device=DEV) * 0.05).requires_grad_(True)
    gamma, beta = (torch.rand(d, device=DEV) + 0.5), torch.randn(d, device=DEV) * 0.1
    h = torch.randn(M, hid, device=DEV).to(dtype).requires_grad_(True)
    res = torch.randn(M, d, device=DEV).to(dtype).requires_grad_(True)
    g1, g2 = torch.randn(M, d, device=DEV).to(dtype), torch.randn(M, d, device=DEV).to(dtype)
    HF.dropout_begin_step()
    taken = []
    orig_take = HF._take_side_v

    def spy(dz, Bt, scale):
        V = orig_take(dz, Bt, scale)
        if V is not None:
            taken.append((dz.detach().clone(), Bt.detach().clone(), float(scale), V.detach().clone()))
        return V

    def run(side: bool):
        HF.LN_BWD_SIDE = side
        HF._DROPOUT["site"] = 7
        t0 = dict(HF.HANDS_TAKEN)
        h.grad = res.grad = None
        gA = torch.zeros_like(A)
        gB = torch.zeros_like(Bm)
        A.grad, Bm.grad = gA, gB
        x = HF.lora_linear(h, pack, A, Bm, 2.0, residual=res, drop_p=lora_p, out_drop_p=0.1)
        xr, xn = HF.layernorm_fork(x, gamma, beta, 1e-5)
        torch.autograd.backward([xr, xn], [g1, g2])
        torch.cuda.synchronize()
        took = {k: HF.HANDS_TAKEN.get(k, 0) - t0.get(k, 0) for k in (HF._H_PRE_MASKED, HF._H_PRE_V)}
        assert took == {HF._H_PRE_MASKED: 1, HF._H_PRE_V: 1 if side else 0}, took       # both rode on the gradient tensors
        return h.grad.clone(), res.grad.clone(), A.grad.clone(), Bm.grad.clone()
    HF._take_side_v = spy
    side_default = HF.LN_BWD_SIDE
    try:
        a = run(True)
        assert len(taken) == 1                               # the adapter's backward took the parked product
        n_on = len(taken)
        b = run(False)
        assert len(taken) == n_on                            # ... and none when the hand-over is off
    finally:
        HF._take_side_v = orig_take
        HF.LN_BWD_SIDE = side_default
    dz, Bt, sc, V = taken[0]
    ref = sc * (dz.float() @ Bt.float().t())
    assert V.shape == (M, r)
    assert (V.float() - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item()
    assert torch.equal(a[1], b[1])                           # the residual gradient does not pass through the adapter
    for u, w in zip(a, b):
        assert rel(u, w) < 4e-3


def test_encoder_train_mode_applies_dropout(tiny_meta=None):
    """RelPosEncoder in .train(): output differs from eval, differs between steps, p = 0 reproduces eval exactly."""
    HF = HFmod()
    from cosyvoice_lora_finetune_framework_amd.modules import Numerics, RelPosEncoder
    torch.manual_seed(0)
    enc = RelPosEncoder(24, output_size=128, attention_heads=2, linear_units=256, num_blocks=2, dropout_rate=0.1,
                        positional_dropout_rate=0.1, attention_dropout_rate=0.0, kind="conformer").to(DEV)
    num = Numerics(dtype=torch.float32)
    B, L = 2, 37
    xs = torch.randn(B * L, 24, device=DEV, requires_grad=True)
    ln = torch.tensor([37, 30], device=DEV, dtype=torch.int32)
    enc.eval()
    y_eval = enc.forward_cl(xs, B, L, ln, num)
    enc.train()
    HF.dropout_begin_step()
    y1 = enc.forward_cl(xs, B, L, ln, num)
    y1.sum().backward()
    assert torch.isfinite(xs.grad).all()
    HF.dropout_begin_step()
    y2 = enc.forward_cl(xs, B, L, ln, num)
    assert rel(y1, y_eval) > 1e-2 and rel(y1, y2) > 1e-2
    for m in enc.modules():
        if hasattr(m, "dropout_rate"):
            m.dropout_rate = 0.0
    enc.embed.out[2].p = 0.0
    assert rel(enc.forward_cl(xs, B, L, ln, num), y_eval) < 1e-6


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("act", [None, "silu"])
def test_lora_dropout_side_path_matches_torch(dtype, act):
    """lora.py:64-76 in train mode: y = act(x W^T + b + s * (drop(x) A^T) B^T) + residual, with a FIXED dropout mask so
    that the HIP path (main GEMM + LoraSideFn) can be compared with torch autograd, forward and all gradients."""
    from cosyvoice_lora_finetune_framework_amd.lora import LoRALinear
    from cosyvoice_lora_finetune_framework_amd.modules import hip_linear
    HF = HFmod()
    torch.manual_seed(11)
    M, K, N = 200, 64, 96
    mod = LoRALinear(torch.nn.Linear(K, N), r=16, lora_alpha=32, lora_dropout=0.25).to(DEV)
    torch.nn.init.normal_(mod.lora_B, std=0.1)
    mask = (torch.rand(M, K, device=DEV) > 0.25).float() / 0.75

    class Fixed(torch.nn.Dropout):
        def forward(self, x):
            return x * mask.to(x.dtype)
    mod.lora_dropout = Fixed(0.25)
    mod.train()
    x = (torch.randn(M, K, device=DEV) * 0.5).to(dtype).requires_grad_(True)
    res = torch.randn(M, N, device=DEV).to(dtype).requires_grad_(True)
    gy = torch.randn(M, N, device=DEV).to(dtype)
    y = hip_linear(mod, x, act=act, residual=res)
    y.backward(gy)
    got = [y.float(), x.grad.float(), res.grad.float(), mod.lora_A.grad.clone(), mod.lora_B.grad.clone()]
    for t in (x, res, mod.lora_A, mod.lora_B):
        t.grad = None
    xf, rf = x.detach().float().requires_grad_(True), res.detach().float().requires_grad_(True)
    W, b = mod.original_layer.weight.float(), mod.original_layer.bias.float()
    z = xf @ W.t() + b + mod.scaling * ((xf * mask) @ mod.lora_A.t()) @ mod.lora_B.t()
    yr = (torch.nn.functional.silu(z) if act else z) + rf
    yr.backward(gy.float())
    ref = [yr, xf.grad, rf.grad, mod.lora_A.grad, mod.lora_B.grad]
    tol = 3e-2 if dtype == torch.bfloat16 else 1e-4
    for name, u, w in zip(("y", "dx", "dres", "dA", "dB"), got, ref):
        assert rel(u, w) < tol, (name, rel(u, w))


def _attn_keep_scale_host(seed: int, site: int, B: int, H: int, L: int, p: float) -> torch.Tensor:
    """Host replica of attention.hip attn_keep_scale / attn_drop_key (SplitMix64 finaliser) -> [B, H, L, L] of {0, 1/(1-p)}."""
    import numpy as np
    M = np.uint64(0xFFFFFFFFFFFFFFFF)

    def mix(z):
        z = (z + np.uint64(0x9e3779b97f4a7c15)) & M
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xbf58476d1ce4e5b9)) & M
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94d049bb133111eb)) & M
        return z ^ (z >> np.uint64(31))
    with np.errstate(over="ignore"):
        key = mix(np.uint64(seed) ^ (np.uint64(site) << np.uint64(32)))
        idx = np.arange(B * H * L * L, dtype=np.uint64)
        u = (mix(key + idx) & np.uint64(0xFFFFFFFF)).astype(np.uint64)
    thr = min(4294967295.0, float(np.float32(p) * np.float32(4294967296.0)))
    keep = u >= np.uint64(int(thr))
    return torch.from_numpy(keep.reshape(B, H, L, L).astype(np.float64)) / (1.0 - float(np.float32(p)))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("causal", [False, True])
def test_attn_relpos_probability_dropout(dtype, causal):
    """attention.py:118 `self.dropout(attn)`: the fused kernels mask the PV operand (not the softmax denominator) with a
    counter-based mask; forward and backward must agree with torch math that uses the SAME mask (replicated on the host)."""
    from oracle import ref_math as R
    HF = HFmod()
    B, H, L, lens, pdrop = 2, 2, 70, [70, 41], 0.1
    d = H * 64
    qkv = q(rnd(B, L, 3 * d, seed=1), dtype)
    p = q(rnd(2 * L - 1, d, seed=2), dtype)
    bu, bv = 0.3 * rnd(H, 64, seed=3), 0.3 * rnd(H, 64, seed=4)
    gy = q(rnd(B, L, d, seed=5), dtype)
    qd = qkv.reshape(B * L, -1).to(DEV, dtype).requires_grad_(True)
    HF.dropout_begin_step()
    o = HF.attn_relpos(qd[:, :d], qd[:, d:2 * d], qd[:, 2 * d:], p.to(DEV, dtype), bu.to(DEV), bv.to(DEV), B, H, L,
                       torch.tensor(lens, dtype=torch.int32, device=DEV), causal, 0.125, dropout_p=pdrop)
    o.backward(gy.reshape(B * L, -1).to(DEV, dtype))
    seed, site = int(HF._DROPOUT["seed"].item()), HF._DROPOUT["site"]
    ks = _attn_keep_scale_host(seed, site, B, H, L, pdrop)
    assert abs(float((ks > 0).double().mean()) - (1 - pdrop)) < 0.02
    qr = qkv.double().requires_grad_(True)
    qq, kk, vv = (t.reshape(B, L, H, 64) for t in qr.split(d, dim=-1))
    pp = p.double().view(1, -1, H, 64).transpose(1, 2)
    ac = (qq + bu.double()).transpose(1, 2) @ kk.transpose(1, 2).transpose(-1, -2)
    bd = R.rel_shift((qq + bv.double()).transpose(1, 2) @ pp.transpose(-1, -2))
    mask = _masks(lens, L).bool().unsqueeze(1)
    if causal:
        mask = mask & torch.tril(torch.ones(L, L, dtype=torch.bool)).unsqueeze(0)
    mm = mask.unsqueeze(1).eq(0)
    at = torch.softmax(((ac + bd) / 8.0).masked_fill(mm, -float("inf")), -1).masked_fill(mm, 0.0)
    orf = ((at * ks) @ vv.transpose(1, 2)).transpose(1, 2).reshape(B, L, d)
    orf.backward(gy.double())
    assert rel(o.reshape(B, L, -1), orf) < TOL[dtype]
    assert rel(qd.grad.reshape(B, L, -1), qr.grad) < TOL[dtype] * 3


def _keep_scale_host(seed: int, site: int, n: int, p: float) -> torch.Tensor:
    """Host replica of common.h cvft_drop_key / cvft_keep4 (tests/helpers.py: keep_fields_host): flat [n] tensor of {0, 1/(1-p)}."""
    import numpy as np
    from helpers import drop_thr_host, keep_fields_host
    u = keep_fields_host(seed, site, (n + 3) // 4).reshape(-1)[:n].astype(np.uint64)
    return torch.from_numpy((u >= np.uint64(drop_thr_host(p))).astype(np.float64)) / (1.0 - float(np.float32(p)))


@pytest.mark.parametrize("r", [16, 64])
@pytest.mark.parametrize("act", [None, "silu"])
def test_lora_dropout_fused_path_matches_torch(act, r):
    """lora.py:64-76 in train mode on the fused bf16 path: the mask lives inside cvft_skinny_dropout (forward) and
    cvft_lora_side_dgrad / cvft_dropout_add (backward); compared with torch math on the host-replicated mask.
    r = 64 (BASELINE configs[4]): ONE adapter whose four rank tiles share the mask site."""
    from cosyvoice_lora_finetune_framework_amd.lora import LoRALinear
    from cosyvoice_lora_finetune_framework_amd.modules import hip_linear
    HF = HFmod()
    torch.manual_seed(5)
    M, K, N, pdrop = 300, 128, 192, 0.15
    mod = LoRALinear(torch.nn.Linear(K, N), r=r, lora_alpha=2 * r, lora_dropout=pdrop).to(DEV)
    torch.nn.init.normal_(mod.lora_B, std=0.1)
    mod.train()
    x = (torch.randn(M, K, device=DEV) * 0.5).to(torch.bfloat16).requires_grad_(True)
    res = torch.randn(M, N, device=DEV).to(torch.bfloat16).requires_grad_(True)
    gy = torch.randn(M, N, device=DEV).to(torch.bfloat16)
    HF.dropout_begin_step()
    y = hip_linear(mod, x, act=act, residual=res)
    assert "LinearFn" in type(y.grad_fn).__name__                       # the fused path, not LoraSideFn
    y.backward(gy)
    seed, site = int(HF._DROPOUT["seed"].item()), HF._DROPOUT["site"]
    mask = _keep_scale_host(seed, site, M * K, pdrop).reshape(M, K).to(DEV).float()
    assert abs(float((mask > 0).float().mean()) - (1 - pdrop)) < 0.02
    got = [y.float(), x.grad.float(), res.grad.float(), mod.lora_A.grad.clone(), mod.lora_B.grad.clone()]
    for t in (mod.lora_A, mod.lora_B):
        t.grad = None
    xf, rf = x.detach().float().requires_grad_(True), res.detach().float().requires_grad_(True)
    W, b = mod.original_layer.weight.float(), mod.original_layer.bias.float()
    z = xf @ W.t() + b + mod.scaling * ((xf * mask) @ mod.lora_A.t()) @ mod.lora_B.t()
    yr = (torch.nn.functional.silu(z) if act else z) + rf
    yr.backward(gy.float())
    for name, u, w in zip(("y", "dx", "dres", "dA", "dB"), got, [yr, xf.grad, rf.grad, mod.lora_A.grad, mod.lora_B.grad]):
        assert rel(u, w) < 3e-2, (name, rel(u, w))


@pytest.mark.parametrize("defer", [False, True])
def test_qkv_stacked_lora_dropout_matches_torch(defer):
    """Stacked q|k|v under lora_dropout: three mask sites in one cvft_skinny_dropout launch, one side-dgrad launch,
    dA of the three adapters from three re-derived dropped inputs in one multi-problem slab launch -- in line, or
    (defer, the default) postponed to the sink's end-of-backward batch launches."""
    from cosyvoice_lora_finetune_framework_amd.lora import LoRALinear
    from cosyvoice_lora_finetune_framework_amd.modules import hip_qkv
    from cosyvoice_lora_finetune_framework_amd.optim import FlatAdamW
    HF = HFmod()
    torch.manual_seed(3)
    K, N, M, pdrop = 256, 512, 1000, 0.05
    mods = [LoRALinear(torch.nn.Linear(K, N), r=16, lora_alpha=32, lora_dropout=pdrop).to(DEV) for _ in range(3)]
    for m in mods:
        torch.nn.init.normal_(m.lora_B, std=0.05)
        m.train()
    params = [p for m in mods for p in (m.lora_A, m.lora_B)]
    opt = FlatAdamW(params, lr=1e-3)
    x = (torch.randn(M, K, device=DEV) * 0.5).to(torch.bfloat16).requires_grad_(True)
    g = [torch.randn(M, N, device=DEV).to(torch.bfloat16) for _ in range(3)]
    opt.zero_grad()
    HF.dropout_begin_step()
    saved, HF.STACKED_DROP_DEFER = HF.STACKED_DROP_DEFER, defer
    try:
        with HF.LoraGradSink() as sink:
            q_, k_, v_ = hip_qkv(mods[0], mods[1], mods[2], x)
            assert "Stacked" in type(q_.grad_fn).__name__
            fused = torch.cat(g, 1)
            torch.autograd.backward([q_, k_, v_], [fused[:, :N], fused[:, N:2 * N], fused[:, 2 * N:]])
            assert bool(sink.deferred) == defer
    finally:
        HF.STACKED_DROP_DEFER = saved
    seed, last = int(HF._DROPOUT["seed"].item()), HF._DROPOUT["site"]
    xf = x.detach().float().requires_grad_(True)
    outs = []
    for i, m in enumerate(mods):
        mask = _keep_scale_host(seed, last - 2 + i, M * K, pdrop).reshape(M, K).to(DEV).float()
        W, b = m.original_layer.weight.float(), m.original_layer.bias.float()
        A, Bm = m.lora_A.detach().clone().requires_grad_(True), m.lora_B.detach().clone().requires_grad_(True)
        y = xf @ W.t() + b + m.scaling * ((xf * mask) @ A.t()) @ Bm.t()
        outs.append((y, A, Bm))
    torch.autograd.backward([o[0] for o in outs], [t.float() for t in g])
    for i, (y, A, Bm) in enumerate(outs):
        assert rel((q_, k_, v_)[i].float(), y) < 2e-2
        assert rel(mods[i].lora_A.grad, A.grad) < 3e-2, ("dA", i, rel(mods[i].lora_A.grad, A.grad))
        assert rel(mods[i].lora_B.grad, Bm.grad) < 3e-2, ("dB", i)
    assert rel(x.grad.float(), xf.grad) < 3e-2


def test_sink_early_batches_equal_end_of_backward_batches():
    """the postponed adapter-gradient products leave in batches on their chain's stream during backward
    (CVFT_SINK_DEFER_EARLY, the shipped default) or all at flush().  Two chains on two streams run the same stacked q|k|v
    adapters (four postponed products each per call, two calls per chain: eight per chain); with batches of three, each chain
    sends two full batches early and leaves two rows for flush(), where the two chains' leftovers meet in one launch.  The
    adapter gradients equal the all-at-flush() form bit for bit (same slabs, same reduce order)."""
    from cosyvoice_lora_finetune_framework_amd.lora import LoRALinear
    from cosyvoice_lora_finetune_framework_amd.modules import hip_qkv
    from cosyvoice_lora_finetune_framework_amd.optim import FlatAdamW
    HF = HFmod()
    torch.manual_seed(5)
    K, N, pdrop = 256, 512, 0.05
    mods = [LoRALinear(torch.nn.Linear(K, N), r=16, lora_alpha=32, lora_dropout=pdrop).to(DEV) for _ in range(3)]
    for m in mods:
        torch.nn.init.normal_(m.lora_B, std=0.05)
        m.train()
    params = [p for m in mods for p in (m.lora_A, m.lora_B)]
    opt = FlatAdamW(params, lr=1e-3)
    xs = [(torch.randn(M, K, device=DEV) * 0.5).to(torch.bfloat16) for M in (1000, 700, 520, 333)]
    gs = [torch.randn(x.shape[0], 3 * N, device=DEV).to(torch.bfloat16) for x in xs]
    side = torch.cuda.Stream()
    saved = (HF.SINK_DEFER_EARLY, HF.SINK_EARLY_BATCH, HF.STACKED_DROP_DEFER, HF.LoraGradSink.uses_hint)
    res = []
    try:
        HF.SINK_EARLY_BATCH, HF.STACKED_DROP_DEFER = 3, True
        for early in (False, True):
            HF.SINK_DEFER_EARLY = early
            HF.LoraGradSink.uses_hint = 4
            opt.zero_grad()
            HF.dropout_begin_step()
            HF._DROPOUT["seed"].fill_(12345)
            launches = []
            orig = HF.LoraGradSink._launch_rows
            HF.LoraGradSink._launch_rows = staticmethod(lambda r, rows: (launches.append((len(rows), torch.cuda.current_stream().cuda_stream)), orig(r, rows))[1])
            try:
                with HF.LoraGradSink() as sink:
                    outs = []
                    main = torch.cuda.current_stream()
                    side.wait_stream(main)
                    for ci, st in enumerate((main, side)):                 # chain ci: two calls on its stream
                        with torch.cuda.stream(st):
                            for x, g in zip(xs[2 * ci: 2 * ci + 2], gs[2 * ci: 2 * ci + 2]):
                                xi = x.clone().requires_grad_(True)
                                q_, k_, v_ = hip_qkv(mods[0], mods[1], mods[2], xi)
                                outs.append(((q_, k_, v_), g))
                    for (q_, k_, v_), g in outs:                           # backward runs on each call's forward stream
                        torch.autograd.backward([q_, k_, v_], [g[:, :N], g[:, N:2 * N], g[:, 2 * N:]])
                    main.wait_stream(side)
                    left = sum(len(v) for v in sink.deferred.values())
            finally:
                HF.LoraGradSink._launch_rows = orig
            torch.cuda.synchronize()
            res.append(([p.grad.clone() for p in params], launches, left))
    finally:
        HF.SINK_DEFER_EARLY, HF.SINK_EARLY_BATCH, HF.STACKED_DROP_DEFER, HF.LoraGradSink.uses_hint = saved
    (g0, l0, left0), (g1, l1, left1) = res
    print("[sink batches] at flush only:", l0, "left", left0, "| early:", l1, "left", left1)
    assert left0 == 16 and sum(n for n, _ in l0) == 16                     # everything at flush() (one launch per rank)
    early = [(n, st) for n, st in l1 if n == 3]
    assert len(early) >= 2 and len({st for _, st in early}) == 2           # full batches left early, each on its own chain's stream
    assert 0 < left1 < 16 and sum(n for n, _ in l1) == 16                  # ... and the leftovers of both chains with flush()
    for a, b in zip(g0, g1):
        assert float(a.abs().max()) > 0 and torch.equal(a, b)


@pytest.mark.parametrize("M,K,R,nsites", [(4000, 256, 48, 3), (5376, 1024, 48, 3), (5376, 1024, 16, 1), (1000, 512, 16, 1),
                                          (333, 320, 16, 1), (77, 64, 48, 3)])
def test_layernorm_with_rank_side_product(M, K, R, nsites):
    """cvft_ln_skinny_dropout: LayerNorm and the dropped rank-side product of its output in one launch -- against the two
    launches it replaces (cvft_layernorm_fwd, then cvft_skinny_dropout on the same y with the same mask sites).  Covers the
    1 / 2 / 4 k-step instantiations, a K whose last waves have an empty slice (320), a partial last row block."""
    HF = HFmod()
    g = torch.Generator().manual_seed(M + K)
    x = (torch.randn(M, K, generator=g) * 2.0 + 0.5).to(DEV, torch.bfloat16)
    gamma = (1.0 + 0.2 * torch.randn(K, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(K, generator=g)).to(DEV)
    A = (torch.randn(R, K, generator=g) / math.sqrt(K)).to(DEV, torch.bfloat16)
    alpha, pdrop, eps = 2.0, 0.15, 1e-5
    HF.dropout_begin_step()
    assert HF.can_ln_skinny(x, A, gamma, beta)
    y, mean, rstd = HF.ln_skinny_dropout(x, gamma, beta, eps, A, alpha, pdrop, nsites)
    U, sites = HF.take_pre_u(y, A, alpha, pdrop, nsites)
    xd = [HF._DROPPED.pop(st) for st in sites]
    # LayerNorm part: the stand-alone kernel (different reduction order: allow one bf16 ulp on y)
    y_ref = HF.layernorm(x, gamma, beta, eps)
    xf = x.float()
    mu_ref = xf.mean(1)
    rs_ref = 1.0 / torch.sqrt(xf.var(1, unbiased=False) + eps)
    assert rel(mean, mu_ref) < 1e-5 and rel(rstd, rs_ref) < 1e-5
    assert (y.float() - y_ref.float()).abs().max() <= 2.0 ** -7 * y_ref.float().abs().max()
    assert (y != y_ref).float().mean() < 0.02
    # rank-side part: bit-identical to the stand-alone kernel on the same y and mask sites
    U_ref = HF.skinny_dropout(y, A, alpha, pdrop, list(sites), keep_dropped=True)
    assert torch.equal(U, U_ref)
    for st, t in zip(sites, xd):
        assert torch.equal(t, HF._DROPPED.pop(st))


@pytest.mark.parametrize("M,N,K,R", [(300, 128, 192, 16), (1000, 256, 1536, 48), (5328, 1024, 1024, 16), (2056, 520, 256, 64)])
def test_gemm_masked_rank_extension(M, N, K, R):
    """cvft_gemm with xdrop (the lora_dropout dgrad inside the GEMM launch): C = A W^T + sum_t mask_t/(1-p) * (U_t Bl_t^T) with
    one mask site per 16-wide rank tile, masks over the output elements -- against torch with the host-replicated masks.
    W is small so that the masked term dominates; covers the 64x64, 128x64 and 128x128 tiles, a partial last tile, a residual."""
    HF = HFmod()
    g = torch.Generator().manual_seed(M + R)
    rn = lambda *s: torch.randn(*s, generator=g)
    x, w = q(rn(M, K), torch.bfloat16), q(rn(N, K) * 0.01, torch.bfloat16)
    u, bl = q(rn(M, R), torch.bfloat16), q(rn(N, R), torch.bfloat16)
    res = q(rn(M, N), torch.bfloat16)
    pdrop, nt = 0.15, R // 16
    HF.dropout_begin_step()
    sites = [HF._next_drop_site() for _ in range(nt)] if R != 64 else [HF._next_drop_site()] * nt
    bd = lambda t: t.to(DEV, torch.bfloat16)
    y = HF.gemm(bd(x), bd(w), U=bd(u), Bl=bd(bl), residual=bd(res), xdrop=(pdrop, sites))
    assert HF.lib().cvft_gemm_last_kernel().decode().endswith(",xdrop")
    seed = int(HF._DROPOUT["seed"].item())
    ref = x.double() @ w.double().t() + res.double()
    for t in range(nt):
        mask = _keep_scale_host(seed, sites[t], M * N, pdrop).reshape(M, N)
        ref = ref + mask * (u[:, 16 * t:16 * t + 16].double() @ bl[:, 16 * t:16 * t + 16].double().t())
    assert rel(y, ref) < 6e-3, rel(y, ref)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("act,n", [("silu", 4096 * 3), ("relu", 1001)])
def test_act_dropout_one_pass(dtype, act, n):
    """cvft_act_dropout: h = dropout(act(z)) and dz = keep/(1-p) * dh * act'(z), each one pass, against torch on the
    host-replicated mask (vector path n % 4 == 0 and the scalar tail path)."""
    HF = HFmod()
    pdrop = 0.1
    z = q(rnd(n, seed=1) * 2, dtype)
    gh = q(rnd(n, seed=2), dtype)
    zd = z.to(DEV, dtype).requires_grad_(True)
    HF.dropout_begin_step()
    h = HF.act_dropout(zd, act, pdrop)
    h.backward(gh.to(DEV, dtype))
    mask = _keep_scale_host(int(HF._DROPOUT["seed"].item()), HF._DROPOUT["site"], n, pdrop)
    zr = z.double().requires_grad_(True)
    hr = (F.silu(zr) if act == "silu" else F.relu(zr)) * mask
    hr.backward(gh.double())
    assert rel(h, hr) < TOL[dtype] and rel(zd.grad, zr.grad) < TOL[dtype] * 2


def test_fp8_mfma_operand_layout_probe():
    """v_mfma_scale_f32_16x16x128_f8f6f4 (unit e8m0 scales, both operands OCP e4m3): lane l supplies row l & 15, bytes
    32 (l >> 4) .. + 31 -- exact on exactly-representable data (csrc/gemm_fp8.hip; groundwork of BASELINE configs[4])."""
    from cosyvoice_lora_finetune_framework_amd.hipops import binding as cb
    g = torch.Generator().manual_seed(0)
    vals = torch.tensor([-4., -3., -2., -1.5, -1., -.5, 0., .5, 1., 1.5, 2., 3., 4., 6., 8., .25])
    A, B = vals[torch.randint(0, 16, (16, 128), generator=g)], vals[torch.randint(0, 16, (16, 128), generator=g)]
    ad, bd = A.to(torch.float8_e4m3fn).view(torch.uint8).to(DEV), B.to(torch.float8_e4m3fn).view(torch.uint8).to(DEV)
    Cd = torch.zeros(16, 16, device=DEV)
    cb.check(cb.lib().cvft_debug_mfma_fp8_probe(ad.data_ptr(), bd.data_ptr(), Cd.data_ptr(), None), "probe")
    assert torch.equal(Cd.cpu().double(), A.double() @ B.double().t())


@pytest.mark.parametrize("M,N,K,R", [(300, 192, 256, 16), (1024, 512, 1024, 0), (5328, 1024, 1024, 16), (2056, 520, 384, 64)])
def test_fp8_quant_and_gemm(M, N, K, R):
    """cvft_quant_fp8_rows against torch's e4m3 conversion of the same scaled rows (bit-exact), and cvft_gemm_fp8 against the
    fp64 product of the DEQUANTISED operands (+ LoRA term, bias, residual): the kernel's own error is fp32 accumulation only;
    the quantisation error against the unquantised product is reported by the looser second check."""
    HF = HFmod()
    g = torch.Generator().manual_seed(M + N)
    rn = lambda *s: torch.randn(*s, generator=g)
    bd = lambda t: t.to(DEV, torch.bfloat16)
    x, w = bd(rn(M, K)), bd(rn(N, K) / K ** 0.5)
    xq, xs = HF.quant_fp8_rows(x)
    wq, ws = HF.quant_fp8_rows(w)
    xf = x.float().cpu()
    s_ref = xf.abs().amax(dim=1) * (1.0 / 448.0)
    assert torch.equal(xs.cpu(), s_ref)
    q_ref = (xf * (1.0 / s_ref).unsqueeze(1)).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(xq.cpu(), q_ref)
    u, bl = (bd(rn(M, R)), bd(rn(N, R) * 0.1)) if R else (None, None)
    b, res = rn(N).to(DEV), bd(rn(M, N))
    y = HF.gemm_fp8(xq, xs, wq, ws, bias=b, U=u, Bl=bl, residual=res)
    deq = lambda qt, sc: qt.cpu().view(torch.float8_e4m3fn).double() * sc.cpu().double().unsqueeze(1)
    ref = deq(xq, xs) @ deq(wq, ws).t() + b.cpu().double() + res.cpu().double()
    true = x.cpu().double() @ w.cpu().double().t() + b.cpu().double() + res.cpu().double()
    if R:
        ext = u.cpu().double() @ bl.cpu().double().t()
        ref, true = ref + ext, true + ext
    assert rel(y, ref) < 4e-3, rel(y, ref)                      # bf16 output rounding
    assert rel(y, true) < 6e-2, rel(y, true)                    # e4m3 operands: a few per cent


def test_fp8_linear_path_close_to_bf16():
    """The opt-in fp8 arithmetic of the frozen-W GEMMs (HF.FP8_ON; BASELINE configs[4]) through LinearFn: forward, dgrad and the
    LoRA gradients stay within e4m3 quantisation error of the bf16 path (the LoRA term itself is not quantised)."""
    from cosyvoice_lora_finetune_framework_amd.lora import LoRALinear
    from cosyvoice_lora_finetune_framework_amd.modules import hip_linear
    HF = HFmod()
    torch.manual_seed(11)
    M, K, N = 4352, 512, 768                    # M * N above the in-launch side-path limit: the plain (U, Bl) main GEMM
    mod = LoRALinear(torch.nn.Linear(K, N), r=16, lora_alpha=32, lora_dropout=0.0).to(DEV).eval()
    torch.nn.init.normal_(mod.lora_B, std=0.05)
    x0 = (torch.randn(M, K, device=DEV) * 0.7).to(torch.bfloat16)
    gy = torch.randn(M, N, device=DEV).to(torch.bfloat16)
    outs = []
    old = (HF.FP8_ON, HF.FP8_MIN_WORK)
    try:
        for on in (False, True):
            HF.FP8_ON, HF.FP8_MIN_WORK = on, 0
            x = x0.clone().requires_grad_(True)
            for t in (mod.lora_A, mod.lora_B):
                t.grad = None
            y = hip_linear(mod, x, act="silu")
            if on:
                assert HF.lib().cvft_gemm_last_kernel().decode().startswith("gemm_fp8_kernel")
            y.backward(gy)
            outs.append([y.float(), x.grad.float(), mod.lora_A.grad.clone(), mod.lora_B.grad.clone()])
    finally:
        HF.FP8_ON, HF.FP8_MIN_WORK = old
    for name, a, b in zip(("y", "dx", "dA", "dB"), outs[1], outs[0]):
        assert rel(a, b) < 8e-2, (name, rel(a, b))
    assert rel(outs[1][0], outs[0][0]) > 1e-4            # and it really ran in different arithmetic


@pytest.mark.parametrize("rpb", [64, 256, 1024])
@pytest.mark.parametrize("r", [16, 48])
def test_lora_rank_partial_batch_matches_torch(r, rpb):
    """cvft_lora_rank_partial_batch: slab products of MANY layers (each its own row count / width / orientation) in one
    launch (70 problems -> two launches of <= 64), summed over their slabs and compared with torch.  rpb 64: one wave per
    64-column stripe and slab; 256 / 1024 (multiples of 128): the stacked form, four waves per stripe on a quarter of the slab's
    rows each -- including slabs shorter than a quarter (M = 64: three waves only join the reduction) and ragged last slabs."""
    import ctypes
    from cosyvoice_lora_finetune_framework_amd.hipops import binding as cb
    HF = HFmod()
    g = torch.Generator().manual_seed(r)
    probs, keep, refs = [], [], []
    shapes = [(1000, 256, 0), (777, 768, 1), (4000, 512, 0), (64, 64, 1), (2056, 1024, 0)] * 14
    arr = (cb.RankProbM * len(shapes))()
    for e, (M, Cn, tr) in zip(arr, shapes):
        Wd = torch.randn(M, Cn, generator=g).to(DEV, torch.bfloat16)
        Rk = torch.randn(M, r, generator=g).to(DEV, torch.bfloat16)
        ns = -(-M // rpb)
        ws = torch.zeros(ns * r * Cn, dtype=torch.float32, device=DEV)
        e.M, e.C, e.Wd, e.ldw, e.Rk, e.ldr, e.part, e.transpose_out, e.rows_per_block = M, Cn, Wd.data_ptr(), Cn, Rk.data_ptr(), r, ws.data_ptr(), tr, rpb
        keep.append((Wd, Rk, ws, ns, tr, Cn))
    cb.check(cb.lib().cvft_lora_rank_partial_batch(r, len(shapes), arr, None), "cvft_lora_rank_partial_batch")
    torch.cuda.synchronize()
    for Wd, Rk, ws, ns, tr, Cn in keep[::7]:
        got = ws.view(ns, Cn, r).sum(0).t() if tr else ws.view(ns, r, Cn).sum(0)
        ref = Rk.double().t() @ Wd.double()
        assert rel(got, ref) < 1e-5, (Wd.shape, tr, rel(got, ref))
