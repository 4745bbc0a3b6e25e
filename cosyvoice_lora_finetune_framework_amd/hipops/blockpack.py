"""MFMA A-operand images of frozen weights for the row-tile chain kernels (csrc/block_fused.hip; layouts: include/cvft.h,
"Estimator transformer block as row-tile chain kernels").

A fragment is what ONE ``v_mfma_f32_32x32x16_bf16`` takes as its A operand: 64 lanes x 8 bf16, lane l = 32 h + r holding
``Wm[32 rt + r][k(ks, h, j)]``, stored as 1 KB contiguous so that a wave streams it with one 16-byte load per lane.  The
index tables are built once per shape with torch integer arithmetic; packing a weight is one gather.
"""
from __future__ import annotations

import functools

import torch


@functools.lru_cache(maxsize=None)
def _index(kind: str, rows: int, K: int) -> torch.Tensor:
    """source element indices of the fragment images of a row-major [rows][K] matrix:
    natural -> [rt][ks][lane][j], chained -> [kt][rt][s][lane][j]"""
    assert rows % 32 == 0 and K % 32 == 0
    lane = torch.arange(64)
    r, h = (lane & 31).view(1, 1, 64, 1), (lane >> 5).view(1, 1, 64, 1)
    j = torch.arange(8).view(1, 1, 1, 8)
    rt = torch.arange(rows // 32)
    if kind == "natural":
        ks = torch.arange(K // 16)
        src = (32 * rt.view(-1, 1, 1, 1) + r) * K + 16 * ks.view(1, -1, 1, 1) + 8 * h + j            # [rt][ks][lane][j]
        return src.contiguous()
    if kind == "chained":
        kt = torch.arange(K // 32).view(-1, 1, 1, 1, 1)
        s = torch.arange(2).view(1, 1, 2, 1, 1)
        r5, h5, j5 = r.view(1, 1, 1, 64, 1), h.view(1, 1, 1, 64, 1), j.view(1, 1, 1, 1, 8)
        col = 32 * kt + 16 * s + 8 * (j5 >> 2) + 4 * h5 + (j5 & 3)
        src = (32 * rt.view(1, -1, 1, 1, 1) + r5) * K + col                                           # [kt][rt][s][lane][j]
        return src.contiguous()
    raise ValueError(kind)


def pack_a(Wm: torch.Tensor, kind: str) -> torch.Tensor:
    """bf16 fragment image of the row-major matrix Wm [rows][K]: natural -> [rt][ks][64][8], chained -> [kt][rt][s][64][8]."""
    rows, K = Wm.shape
    idx = _index(kind, rows, K).to(Wm.device)
    return Wm.detach().to(torch.bfloat16).contiguous().view(-1)[idx]


def _streams(chunks_per_wave):
    """[4 waves][wave_frags][64][8] from per-wave lists of [n_i][64][8] fragment blocks, + 32 fragments of padding"""
    waves = [torch.cat([c.reshape(-1, 64, 8) for c in chunks], 0) for chunks in chunks_per_wave]
    assert len({w.shape[0] for w in waves}) == 1
    pad = waves[0].new_zeros((32, 64, 8))
    return torch.cat(waves + [pad], 0).contiguous(), waves[0].shape[0]


class BlockTailPack:
    """Frozen weights of one estimator block's tail -- to_out (optional), norm3, ff.net[0].proj, ff.net[2] -- as the weight
    streams of cvft_block_tail_fwd / _bwd (include/cvft.h: each wave's fragments in the order it consumes them)."""

    def __init__(self, w_out, b_out, gamma, beta, eps: float, w1, b1, w2, b2):
        dev = w1.device
        f32 = lambda t, n: (torch.zeros(n, device=dev) if t is None else t.detach().float()).contiguous()
        self.F, self.D = w1.shape
        assert self.D == 256 and self.F % 128 == 0 and self.F <= 2048 and tuple(w2.shape) == (self.D, self.F)
        self.eps = float(eps)
        self.gamma, self.beta = f32(gamma, self.D), f32(beta, self.D)
        self.b1, self.b2 = f32(b1, self.F), f32(b2, self.D)
        self.DI = 0 if w_out is None else w_out.shape[1]
        assert self.DI in (0, 256, 512) and (w_out is None or w_out.shape[0] == self.D)
        self.bo = None if w_out is None else f32(b_out, self.D)
        n = self.F // 128
        W1n = pack_a(w1, "natural")                              # [F/32][16]
        W2c = pack_a(w2, "chained")                              # [F/32 kt][8 ct][2 s]
        W2Tn = pack_a(w2.detach().t().contiguous(), "natural")   # [F/32][16]
        W1Tc = pack_a(w1.detach().t().contiguous(), "chained")   # [F/32 kt][8 dt][2 s]
        if self.DI:
            Won = pack_a(w_out, "natural")                       # [8 ct][DI/16 ks]
            WoTn = pack_a(w_out.detach().t().contiguous(), "natural")   # [DI/32 ft][16 ks]
            q = self.DI // 64
        fwd, bwd = [], []
        for w in range(4):
            cf, cb = [], []
            if self.DI:
                cf.append(Won[:, q * w:q * (w + 1)].permute(1, 0, 2, 3))             # [ks][ct]
            cf.append(W1n[n * w])
            cb.append(W2Tn[n * w])
            for t in range(n):
                ht = n * w + t
                if t + 1 < n:
                    cf.append(W1n[ht + 1])
                    cb.append(W2Tn[ht + 1])
                cf.append(W2c[ht].permute(1, 0, 2, 3))                                # [s][ct]
                cb.append(W1Tc[ht].permute(1, 0, 2, 3))                               # [s][dt]
            if self.DI:
                fw = self.DI // 128                                                   # feature tiles per wave
                blk = WoTn[fw * w:fw * (w + 1)].reshape(fw // 2, 2, 16, 64, 8)        # [r][f2][ks]
                cb.append(blk.permute(0, 2, 1, 3, 4))                                 # [r][ks][f2]
            fwd.append(cf)
            bwd.append(cb)
        self.W_fwd, nf = _streams(fwd)
        self.W_bwd, nb = _streams(bwd)
        assert nf == self.DI // 8 + self.F // 4 and nb == nf
        # "lean" streams (csrc/block_lean.hip): a wave owns 64 OUTPUT features of every link and hidden tile 4 r + w of round r
        W2n = pack_a(w2, "natural")                              # [8 ct][F/16 ks]
        W1Tn = pack_a(w1.detach().t().contiguous(), "natural")   # [8 dt][F/16 ks]
        nr = self.F // 128
        fwd, bwd = [], []
        for w in range(4):
            cf, cb = [], []
            if self.DI:
                cf.append(Won[2 * w:2 * w + 2].permute(1, 0, 2, 3))                   # [ks][c2]
            cf.append(W1n[w])
            cb.append(W2Tn[w])
            for r in range(nr):
                if r + 1 < nr:
                    cf.append(W1n[4 * (r + 1) + w])
                    cb.append(W2Tn[4 * (r + 1) + w])
                cf.append(W2n[2 * w:2 * w + 2, 8 * r:8 * r + 8].permute(1, 0, 2, 3))     # [k'][c2]
                cb.append(W1Tn[2 * w:2 * w + 2, 8 * r:8 * r + 8].permute(1, 0, 2, 3))
            if self.DI:
                fw = self.DI // 128
                cb.append(WoTn[fw * w:fw * (w + 1)].permute(1, 0, 2, 3))                # [ks][f]
            fwd.append(cf)
            bwd.append(cb)
        self.W_fwd_lean, nfl = _streams(fwd)
        self.W_bwd_lean, nbl = _streams(bwd)
        assert nfl == nf and nbl == nf
        # "wide" streams (csrc/block_wide.hip): the lean groups with the second product lagging one round:
        # G1(0), G1(1), { G1(r + 1), G2(r - 1) : r = 1 .. nr - 2 }, G2(nr - 2), G2(nr - 1)      (nr = 1: G1(0), G2(0))
        order = [("1", 0)] + ([("1", 1)] if nr > 1 else [])
        for r in range(1, nr - 1):
            order += [("1", r + 1), ("2", r - 1)]
        order += ([("2", nr - 2)] if nr > 1 else []) + [("2", nr - 1)]
        assert sorted(order) == sorted([(k, r) for k in "12" for r in range(nr)])
        fwd, bwd = [], []
        for w in range(4):
            g1f = lambda r: W1n[4 * r + w]
            g2f = lambda r: W2n[2 * w:2 * w + 2, 8 * r:8 * r + 8].permute(1, 0, 2, 3)          # [k'][c2]
            g1b = lambda r: W2Tn[4 * r + w]
            g2b = lambda r: W1Tn[2 * w:2 * w + 2, 8 * r:8 * r + 8].permute(1, 0, 2, 3)
            cf = [Won[2 * w:2 * w + 2].permute(1, 0, 2, 3)] if self.DI else []
            cf += [(g1f if k == "1" else g2f)(r) for k, r in order]
            cb = [(g1b if k == "1" else g2b)(r) for k, r in order]
            if self.DI:
                fw = self.DI // 128
                cb.append(WoTn[fw * w:fw * (w + 1)].permute(1, 0, 2, 3))                        # [ks][f]
            fwd.append(cf)
            bwd.append(cb)
        self.W_fwd_wide, nfw = _streams(fwd)
        self.W_bwd_wide, nbw = _streams(bwd)
        assert nfw == nf and nbw == nf
        # eight-wave 64-row forward (csrc/block_wide8.hip): wave w of 8 owns feature tile w of every link and hidden tile 8 r + w of round r
        # (rounds of 256 hidden units), groups of 16 fragments in the same lagged order
        self.W_fwd_wide8 = None
        if self.F % 256 == 0 and 512 <= self.F <= 1024:
            nr8 = self.F // 256
            order8 = [("1", 0), ("1", 1)]
            for r in range(1, nr8 - 1):
                order8 += [("1", r + 1), ("2", r - 1)]
            order8 += [("2", nr8 - 2), ("2", nr8 - 1)]
            assert sorted(order8) == sorted([(k, r) for k in "12" for r in range(nr8)])
            fwd = []
            for w in range(8):
                cf = [Won[w]] if self.DI else []                                             # [ks]
                cf += [(W1n[8 * r + w] if k == "1" else W2n[w, 16 * r:16 * r + 16]) for k, r in order8]
                fwd.append(cf)
            self.W_fwd_wide8, nf8 = _streams(fwd)
            assert nf8 == self.DI // 16 + self.F // 8


class BlockQkvPack:
    """Frozen stacked q|k|v weight [3N][256] of one estimator block (rows of to_q | to_k | to_v) as the weight streams of
    cvft_block_qkv_fwd / _bwd (include/cvft.h), plus the bias and the norm1 affine parameters."""

    def __init__(self, wqkv, bias, gamma, beta, eps: float):
        dev = wqkv.device
        self.N3, self.D = wqkv.shape
        assert self.D == 256 and self.N3 == 1536
        f32 = lambda t, n: (torch.zeros(n, device=dev) if t is None else t.detach().float()).contiguous()
        self.eps = float(eps)
        self.gamma, self.beta = f32(gamma, self.D), f32(beta, self.D)
        self.bias = None if bias is None else f32(bias, self.N3)
        Wn = pack_a(wqkv, "natural")                                 # [48 nt][16 ks]
        WTn = pack_a(wqkv.detach().t().contiguous(), "natural")      # [8 ct][96 ks]
        fwd = [[Wn[12 * w:12 * (w + 1)]] for w in range(4)]                                  # [nt][ks]
        bwd = [[WTn[:, 24 * w:24 * (w + 1)].permute(1, 0, 2, 3)] for w in range(4)]          # [ks][ct]
        self.W_fwd, nf = _streams(fwd)
        self.W_bwd, nb = _streams(bwd)
        assert nf == nb == self.N3 // 8
        # 64-row form (csrc/block_qkv_wide.hip, 8 waves): the forward stream is the same; backward, wave w owns feature tile w over all
        # of 3N: [ks]
        bwd = [[WTn[w]] for w in range(8)]
        self.W_bwd_wide, nbw = _streams(bwd)
        assert nbw == self.N3 // 16


class BlockLinkPack:
    """Weight streams of cvft_block_link_fwd / _bwd (include/cvft.h): block i's tail and block i + 1's q|k|v head in one launch on the
    same rows.  Forward, per wave: its tail fragments in tail order (BlockTailPack.W_fwd), then its 192 head fragments
    (BlockQkvPack.W_fwd) -- the fragment ring of the kernel runs from one product into the next without a refill; backward: the
    head's W_bwd fragments, then the tail's."""

    def __init__(self, tail: BlockTailPack, head: BlockQkvPack):
        assert tail.DI == 512 and 256 <= tail.F <= 1024, "cvft_block_link_fwd: DI == 512, 256 <= F <= 1024"
        nt, nh = tail.DI // 8 + tail.F // 4, head.N3 // 8
        t = tail.W_fwd[:4 * nt].view(4, nt, 64, 8)
        h = head.W_fwd[:4 * nh].view(4, nh, 64, 8)
        self.W_fwd, n = _streams([[t[w], h[w]] for w in range(4)])
        assert n == nt + nh
        # backward (cvft_block_link_bwd): the head's backward runs first, then the tail's
        tb = tail.W_bwd[:4 * nt].view(4, nt, 64, 8)
        hb = head.W_bwd[:4 * nh].view(4, nh, 64, 8)
        self.W_bwd, n = _streams([[hb[w], tb[w]] for w in range(4)])
        assert n == nt + nh
        self.tail, self.head = tail, head
