"""MFMA A-operand packing tables (hipops/blockpack.py): pure index arithmetic, checked on the CPU against the lane maps of
v_mfma_f32_32x32x16_bf16 (cdna_hip_programming.md section 3) by emulating the instruction with torch."""
import torch

from cosyvoice_lora_finetune_framework_amd.hipops import blockpack as bp


def _mfma(afrag, bfrag):
    """D[32][32] of one 32x32x16 MFMA from per-lane fragments [64][8]: lane l = 32 h + r holds A[r][8 h + j] / B[8 h + j][r]."""
    A = torch.zeros(32, 16, dtype=torch.float64)
    Bm = torch.zeros(16, 32, dtype=torch.float64)
    for l in range(64):
        r, h = l & 31, l >> 5
        A[r, 8 * h:8 * h + 8] = afrag[l].double()
        Bm[8 * h:8 * h + 8, r] = bfrag[l].double()
    return A @ Bm


def _acc_layout(D):
    """accumulator registers per lane: lane (col = l & 31, h = l >> 5), reg q -> row (q & 3) + 8 (q >> 2) + 4 h"""
    acc = torch.zeros(64, 16, dtype=torch.float64)
    for l in range(64):
        for q in range(16):
            acc[l, q] = D[(q & 3) + 8 * (q >> 2) + 4 * (l >> 5), l & 31]
    return acc


def _bfrag_natural(X, ks):
    """B fragment of X^T for activations X [32 rows][K]: lane (m, h) holds X[m][16 ks + 8 h + j]"""
    f = torch.zeros(64, 8)
    for l in range(64):
        f[l] = X[l & 31, 16 * ks + 8 * (l >> 5):16 * ks + 8 * (l >> 5) + 8]
    return f


def test_natural_and_chained_images_compute_the_chain():
    g = torch.Generator().manual_seed(0)
    K, Fh, Dn = 64, 64, 32
    W1 = torch.randint(-3, 4, (Fh, K), generator=g).float()       # exact small integers: any wrong index is an O(1) error
    W2 = torch.randint(-3, 4, (Dn, Fh), generator=g).float()
    X = torch.randint(-3, 4, (32, K), generator=g).float()
    p1 = bp.pack_a(W1, "natural").float()
    p2 = bp.pack_a(W2, "chained").float()
    out = torch.zeros(Dn, 32, dtype=torch.float64)
    for ht in range(Fh // 32):
        D1 = sum(_mfma(p1[ht, ks], _bfrag_natural(X, ks)) for ks in range(K // 16))          # H^T tile [32 hidden][32 rows]
        assert torch.equal(D1, (W1[32 * ht:32 * ht + 32].double() @ X.double().t()))
        acc = _acc_layout(D1)
        for s in range(2):
            hb = acc[:, 8 * s:8 * s + 8]                                                      # registers 8s .. 8s+7 as the B fragment
            out += _mfma(p2[ht, 0, s], hb)
    assert torch.equal(out, (W2.double() @ (W1.double() @ X.double().t())))


def test_block_tail_streams_hold_every_fragment_once_in_consumption_order():
    g = torch.Generator().manual_seed(2)
    Fh, DI = 256, 256
    w1, w2, wo = torch.randn(Fh, 256, generator=g), torch.randn(256, Fh, generator=g), torch.randn(256, DI, generator=g)
    pk = bp.BlockTailPack(wo, None, None, None, 1e-5, w1, None, w2, None)
    n, nf = Fh // 128, DI // 8 + Fh // 4
    st = pk.W_fwd.float().view(-1, 64, 8)
    assert st.shape[0] == 4 * nf + 32 and float(st[4 * nf:].abs().sum()) == 0.0
    W1n, W2c = bp.pack_a(w1, "natural").float(), bp.pack_a(w2, "chained").float()
    Won = bp.pack_a(wo, "natural").float()
    for w in range(4):
        ws = st[w * nf:(w + 1) * nf]
        q = DI // 64
        for ks in range(q):
            for ct in range(8):
                assert torch.equal(ws[ks * 8 + ct], Won[ct, q * w + ks])
        pos = DI // 8
        assert torch.equal(ws[pos:pos + 16], W1n[n * w])
        pos += 16
        for t in range(n):
            if t + 1 < n:
                assert torch.equal(ws[pos:pos + 16], W1n[n * w + t + 1])
                pos += 16
            for s in range(2):
                for ct in range(8):
                    assert torch.equal(ws[pos + 8 * s + ct], W2c[n * w + t, ct, s])
            pos += 16
        assert pos == nf
    # backward stream: same skeleton on the transposed matrices, then the projection's dgrad fragments [r][ks][f2]
    sb = pk.W_bwd.float().view(-1, 64, 8)
    W2Tn, WoTn = bp.pack_a(w2.t().contiguous(), "natural").float(), bp.pack_a(wo.t().contiguous(), "natural").float()
    for w in range(4):
        ws = sb[w * nf:(w + 1) * nf]
        assert torch.equal(ws[:16], W2Tn[n * w])
        base = Fh // 4
        fw = DI // 128
        for r in range(fw // 2):
            for ks in range(16):
                for f2 in range(2):
                    assert torch.equal(ws[base + 32 * r + 2 * ks + f2], WoTn[fw * w + 2 * r + f2, ks])


def _lagged(nr):
    """consumption order of the 64-row forms: G1(0), G1(1), { G1(r + 1), G2(r - 1) : r = 1 .. nr - 2 }, G2(nr - 2), G2(nr - 1)"""
    order = [("1", 0), ("1", 1)]
    for r in range(1, nr - 1):
        order += [("1", r + 1), ("2", r - 1)]
    return order + [("2", nr - 2), ("2", nr - 1)]


def test_64_row_streams_follow_the_lagged_order():
    """W_fwd_wide / W_bwd_wide (four waves, csrc/block_wide.hip) and W_fwd_wide8 (eight waves, csrc/block_wide8.hip): every group of 16
    fragments restated from the natural images, in the order the kernels' comments give"""
    g = torch.Generator().manual_seed(3)
    Fh, DI = 1024, 512
    w1, w2, wo = torch.randn(Fh, 256, generator=g), torch.randn(256, Fh, generator=g), torch.randn(256, DI, generator=g)
    pk = bp.BlockTailPack(wo, None, None, None, 1e-5, w1, None, w2, None)
    W1n, W2n, Won = bp.pack_a(w1, "natural").float(), bp.pack_a(w2, "natural").float(), bp.pack_a(wo, "natural").float()
    W2Tn, W1Tn = bp.pack_a(w2.t().contiguous(), "natural").float(), bp.pack_a(w1.t().contiguous(), "natural").float()
    WoTn = bp.pack_a(wo.t().contiguous(), "natural").float()
    # four waves: wave w owns feature tiles 2 w, 2 w + 1 and hidden tile 4 r + w of round r (128 hidden units per round)
    nf, nr = DI // 8 + Fh // 4, Fh // 128
    sf, sb = pk.W_fwd_wide.float().view(-1, 64, 8), pk.W_bwd_wide.float().view(-1, 64, 8)
    assert sf.shape[0] == sb.shape[0] == 4 * nf + 32
    for w in range(4):
        ws, wb = sf[w * nf:(w + 1) * nf], sb[w * nf:(w + 1) * nf]
        for ks in range(DI // 16):
            for c2 in range(2):
                assert torch.equal(ws[2 * ks + c2], Won[2 * w + c2, ks])
        pos = DI // 8
        for i, (kind, r) in enumerate(_lagged(nr)):
            a = ws[pos + 16 * i:pos + 16 * i + 16]
            b = wb[16 * i:16 * i + 16]
            if kind == "1":
                assert torch.equal(a, W1n[4 * r + w]) and torch.equal(b, W2Tn[4 * r + w])
            else:
                for k in range(8):
                    for c2 in range(2):
                        assert torch.equal(a[2 * k + c2], W2n[2 * w + c2, 8 * r + k]) and torch.equal(b[2 * k + c2], W1Tn[2 * w + c2, 8 * r + k])
        fw = DI // 128
        for ks in range(16):
            for f in range(fw):
                assert torch.equal(wb[32 * nr + ks * fw + f], WoTn[fw * w + f, ks])
    # eight waves: wave w owns feature tile w and hidden tile 8 r + w of round r (256 hidden units per round)
    nf8, nr8 = DI // 16 + Fh // 8, Fh // 256
    s8 = pk.W_fwd_wide8.float().view(-1, 64, 8)
    assert s8.shape[0] == 8 * nf8 + 32
    for w in range(8):
        ws = s8[w * nf8:(w + 1) * nf8]
        assert torch.equal(ws[:DI // 16], Won[w])
        for i, (kind, r) in enumerate(_lagged(nr8)):
            a = ws[DI // 16 + 16 * i:DI // 16 + 16 * i + 16]
            assert torch.equal(a, W1n[8 * r + w] if kind == "1" else W2n[w, 16 * r:16 * r + 16])
    # F = 128 has a single round: no 64-row streams for the eight-wave form, the four-wave one degenerates to G1(0), G2(0)
    small = bp.BlockTailPack(wo, None, None, None, 1e-5, w1[:128], None, w2[:, :128], None)
    assert small.W_fwd_wide8 is None


def test_qkv_streams_of_the_64_row_backward():
    """BlockQkvPack.W_bwd_wide (csrc/block_qkv_wide.hip): wave w of eight = Wqkv^T fragments of feature tile w over all of 3N, [ks]"""
    g = torch.Generator().manual_seed(4)
    wqkv = torch.randn(1536, 256, generator=g)
    pk = bp.BlockQkvPack(wqkv, None, None, None, 1e-5)
    WTn = bp.pack_a(wqkv.t().contiguous(), "natural").float()                  # [8 ct][96 ks]
    st = pk.W_bwd_wide.float().view(-1, 64, 8)
    assert st.shape[0] == 8 * 96 + 32 and float(st[8 * 96:].abs().sum()) == 0.0
    for w in range(8):
        assert torch.equal(st[96 * w:96 * (w + 1)], WTn[w])
