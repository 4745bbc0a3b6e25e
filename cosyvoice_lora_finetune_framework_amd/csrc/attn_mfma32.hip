// attn_mfma32.hip -- bf16 fused attention (head_dim 64) on v_mfma_f32_32x32x16_bf16 with the score tile kept in
// MFMA accumulator layout from QK^T to the products that consume it (no P / dS round trip through LDS).
//
//   forward, dQ  : a wave owns 32 QUERIES.  S^T = K . Q^T puts the query on the lane and the keys in the 16 accumulator
//                  registers, so max / sum / exp are lane-local (one cross-half exchange) and the bf16-packed
//                  accumulators ARE the B operand of O^T = V^T . P^T  (dQ^T = K^T . dS^T) -- cdna_hip_programming.md
//                  section 3, "An accumulator tile as the next MFMA's operand".  V^T / K^T come from the row-major LDS
//                  tile through the transposing LDS read.
//   dK, dV       : a wave owns 32 KEYS.  S = Q . K^T and dP = dO . V^T put the key on the lane; P and dS feed
//                  dV^T = dO^T . P and dK^T = Q^T . dS as B operands.  The constant row terms of the rel-pos scores,
//                  u.k_j and v.p_m, enter as MFMAs against a fragment whose 32 rows all hold u (v): no second LDS image
//                  of Q, no bias adds on fragments.
//   rel-pos      : bd[i,j] = (q_i+v).p[L-1-i+j] is a row-dependent shift of G = (q+v) P^T; the shift is the one thing
//                  that still crosses LDS (per-wave fp32 window, written 16 B wide in accumulator order, read back at
//                  the skewed column).  dQ's P-term takes dS through the inverse shift (bf16 window, zero outside the
//                  written parallelogram) straight into B-operand registers.
// Generic-T kernels (fp32 parity path, gradient w.r.t. p): attention.hip.
// Replaces (reference): modules.py:253-293 / diffusers Attention; cosyvoice/transformer/attention.py:200-330, 82-127.
#include <stdlib.h>
#include <type_traits>
#include "attn_common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define DEV __device__ __forceinline__
#define LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// Diagnostic build only (-DA32_STAMPS, tools/attn_stamps.py): per-wave cycle sums of the loop segments, written to a
// buffer of their own (cdna_hip_programming.md section 7, "In-kernel stamps").  No stamp executes in the product build.
#ifdef A32_STAMPS
__device__ unsigned long long* g_a32_stamps = nullptr;
extern "C" int cvft_debug_attn_stamps(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_a32_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : -2;
}
#define STAMP_DECL() unsigned long long seg_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tl_ = 0; STAMP_T0()
#define STAMP_T0()                                                                             \
    do {                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl_)::"memory");            \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    } while (0)
#define STAMP(k)                                                                               \
    do {                                                                                       \
        unsigned long long t_;                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        seg_[k] += t_ - tl_;                                                                   \
        tl_ = t_;                                                                              \
    } while (0)
#define STAMP_FLUSH()                                                                          \
    do {                                                                                       \
        if (g_a32_stamps && (threadIdx.x & 63) == 0) {                                         \
            const size_t wv_ = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); \
            for (int k_ = 0; k_ < 12; ++k_) g_a32_stamps[wv_ * 12 + k_] = seg_[k_];            \
        }                                                                                      \
    } while (0)
#else
#define STAMP_DECL() do {} while (0)
#define STAMP(k) do {} while (0)
#define STAMP_FLUSH() do {} while (0)
#endif

namespace a32 {
constexpr int LDK = 72;            // LDS tile row stride (elements): 144 B, conflict-free 16-byte row reads
constexpr int TILE = 64 * LDK;     // one staged 64 x 64 tile
constexpr int SKW = 100;           // forward-skew window row stride (dwords), 96 columns used
constexpr int RSK = 100;           // inverse-skew window row stride (bf16 elements = 200 B), 96 columns used
constexpr int SK2 = 64;            // dK/dV skew window row stride (dwords)
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
constexpr float NEGB2 = -1.0e10f * LOG2E;      // the estimator's additive -1e10 key bias, in log2 units
constexpr float NINF = -__builtin_inff();

DEV f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
DEV f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}
DEV bf16x8 zero8() { bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0}; return z; }
// accumulator register r of lane half h holds tile row rowof(r) + 4h
DEV constexpr int rowof(int r) { return (r & 3) + 8 * (r >> 2); }

// rows r0 .. r0+31 of a row-major LDS tile as the A (or B) operand of k-step s: lane (l31, h) reads 16 B of row l31
DEV bf16x8 ld_row(const bf16_t* img, int r0, int s, int lane) {
    return *reinterpret_cast<const bf16x8*>(img + (r0 + (lane & 31)) * LDK + 16 * s + 8 * (lane >> 5));
}
// the TRANSPOSE of tile rows R0 .. R0+15, columns d0 .. d0+31, as the A operand paired with a packed accumulator tile:
// lane (d = d0 + l31, h) gets element e = row R0 + 8(e>>2) + 4h + (e&3)  (ds_read_b64_tr_b16, cdna_hip_programming.md T10)
DEV bf16x8 ld_tr(const bf16_t* img, int R0, int d0, int lane) {
    typedef __attribute__((address_space(3))) bf16x4 lds_b4;
    const bf16_t* p = img + (R0 + 4 * (lane >> 5) + ((lane >> 2) & 3)) * LDK + d0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)p);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(p + 8 * LDK));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// accumulator registers 8*ks .. 8*ks+7 as the bf16 B operand of k-step ks (rows 16ks .. 16ks+15 of the tile)
template <int KS> DEV bf16x8 pack8(const f32x16& a) {
    bf16x8 f;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (bf16_t)a[8 * KS + e];
    return f;
}
// 16-byte fragment of global row `row` (k-step s, half h).  Rows outside [0, rlim) read the nearest valid row instead of
// branching around the load (hipcc waits vmcnt(0) at the join of a conditional load -- cdna_hip_programming.md section 5,
// trap 4c -- which would serialise every prefetch); every consumer masks those rows to exact zeros (p = 0, dS = 0).
DEV bf16x8 gfrag(const bf16_t* __restrict__ g, int row, int rlim, int ld, int s, int h) {
    row = min(max(row, 0), rlim - 1);
    return *reinterpret_cast<const bf16x8*>(g + (size_t)row * ld + 16 * s + 8 * h);
}
// the row fragments of 32 rows (A or B operand of the 4 k-steps) straight from global / L2, row clamped to [0, rlim)
DEV void wave_frag_load(bf16x8 (&f)[4], const bf16_t* __restrict__ g, int ld, int row, int rlim, int hf) {
    row = min(max(row, 0), rlim - 1);
#pragma unroll
    for (int s = 0; s < 4; ++s) f[s] = *reinterpret_cast<const bf16x8*>(g + (size_t)row * ld + 16 * s + 8 * hf);
}
// split staging of one 64 x 64 tile (issue-early / write-late, T14): unconditional loads with clamped rows.  The staging
// registers are plain native vectors and every load in a tile loop is unconditional (a step past the end re-loads the
// last tile): a conditional load, or an array inside a struct that crosses the branch, sends them through scratch memory
// with a vmcnt(0) each -- the prefetch then is synchronous.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef u32x4 Tile2[2];
// NT = threads of the block: 256 (two 16-byte chunks per thread and tile) or 512 (one)
template <int NT>
DEV void tload(Tile2& t, const bf16_t* __restrict__ g, int ld, int r0, int rlim, int tid) {
#pragma unroll
    for (int i = 0; i < 512 / NT; ++i) {
        const int c = tid + NT * i, row = min(max(r0 + (c >> 3), 0), rlim - 1);
        t[i] = *reinterpret_cast<const u32x4*>(g + (size_t)row * ld + (c & 7) * 8);
    }
}
template <int NT>
DEV void tstore(const Tile2& t, bf16_t* S, int tid) {
#pragma unroll
    for (int i = 0; i < 512 / NT; ++i) {
        const int c = tid + NT * i;
        *reinterpret_cast<u32x4*>(S + (c >> 3) * LDK + (c & 7) * 8) = t[i];
    }
}
// tell the waitcnt pass these fragments have landed (a register use it can see), so their first use inside the tile
// loop does not become an s_waitcnt vmcnt(0) that also drains the prefetch in flight
DEV void touch(const bf16x8 (&f)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) asm volatile("" ::"v"(f[s]));
}
DEV bf16x8 bias_frag(bf16x8 f, const float* __restrict__ bias, int s, int h) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (bf16_t)((float)f[e] + bias[16 * s + 8 * h + e]);
    return r;
}
DEV bf16x8 const_frag(const float* __restrict__ bias, int s, int h) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (bf16_t)bias[16 * s + 8 * h + e];
    return r;
}
// combine a per-lane value with the lane 32 away (the other half of the same query / key)
DEV float xh_max(float v) {
    u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
DEV float xh_sum(float v) {
    u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
DEV float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
// 4 consecutive columns of one output row: 8-byte store
DEV void st4(bf16_t* dst, float a, float b, float c, float d) {
    bf16x4 v = {(bf16_t)a, (bf16_t)b, (bf16_t)c, (bf16_t)d};
    *reinterpret_cast<bf16x4*>(dst) = v;
}
// 1-D grid -> (batch*head, owner block): the owner blocks of one (batch, head) get ids that differ by multiples of 8 --
// the same XCD under round-robin dispatch (a speed assumption only) -- and follow each other, so the K/V (or Q/dO) rows
// they all stream are fetched into that XCD's L2 once instead of once per block
DEV void block_map(int n, int nblk, int nbh, int& bh, int& blk) {
    if ((nbh & 7) == 0) {
        const int x = n & 7, k = n >> 3;
        bh = 8 * (k / nblk) + x;
        blk = k % nblk;
    } else {
        bh = n / nblk;
        blk = n % nblk;
    }
}
struct DropCtx { unsigned long long key; unsigned thr; float inv; };
template <bool DROP> DEV DropCtx drop_ctx(const AP<bf16_t>& p) {
    DropCtx d = {0ull, 0u, 1.f};
    if (DROP) {
        d.key = attn_drop_key(p.seed, p.site);
        d.thr = (unsigned)fminf(4294967295.f, p.drop_p * 4294967296.f);
        d.inv = 1.f / (1.f - p.drop_p);
    }
    return d;
}
}  // namespace a32

// =====================================================================================================================
// forward: block = 4 waves x 32 queries, 64 keys per step
// =====================================================================================================================
namespace a32 {
typedef __attribute__((ext_vector_type(2))) float f32x2;
// S^T = K . Q^T of one 64-key step, raw score units, plus (REL) the rel-pos term brought in through the skew window.
// Band rows mlo(t) .. mlo(t)+95 (mlo(t) = L-1 - iw - 31 + 64t): the first 32 are the previous step's last 32 and arrive
// as accumulators (gcarry); pf0 / pf1 hold the P-row fragments of the other 64 (A operand, straight from global / L2).
template <bool REL>
DEV void fwd_scores(f32x16 (&s)[2], const bf16_t* Kc, const bf16x8 (&qu)[4], const bf16x8 (&qv)[4],
                    const bf16x8 (&pf0)[4], const bf16x8 (&pf1)[4], f32x16& gcarry, float* Gw, int lane) {
    const int l31 = lane & 31, hf = lane >> 5;
    s[0] = zero16();
    s[1] = zero16();
#pragma unroll
    for (int ss = 0; ss < 4; ++ss) {
        s[0] = mfma(ld_row(Kc, 0, ss, lane), qu[ss], s[0]);
        s[1] = mfma(ld_row(Kc, 32, ss, lane), qu[ss], s[1]);
    }
    if (REL) {
        f32x16 g1 = zero16(), g2 = zero16();
#pragma unroll
        for (int ss = 0; ss < 4; ++ss) {
            g1 = mfma(pf0[ss], qv[ss], g1);
            g2 = mfma(pf1[ss], qv[ss], g2);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f32x4 a = {gcarry[4 * e], gcarry[4 * e + 1], gcarry[4 * e + 2], gcarry[4 * e + 3]};
            f32x4 b = {g1[4 * e], g1[4 * e + 1], g1[4 * e + 2], g1[4 * e + 3]};
            f32x4 c = {g2[4 * e], g2[4 * e + 1], g2[4 * e + 2], g2[4 * e + 3]};
            *reinterpret_cast<f32x4*>(Gw + 8 * e + 4 * hf) = a;
            *reinterpret_cast<f32x4*>(Gw + 32 + 8 * e + 4 * hf) = b;
            *reinterpret_cast<f32x4*>(Gw + 64 + 8 * e + 4 * hf) = c;
        }
        gcarry = g2;
        LDS_FENCE();
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] += Gw[(31 - l31) + 32 * kt + rowof(r) + 4 * hf];
        LDS_FENCE();
    }
}
}  // namespace a32

// Loop order (one wave): softmax(t) -> P.V MFMAs (t) -> Q.K^T (+ band) MFMAs of step t+1 -> publish / prefetch -> barrier.
// The score MFMAs of the NEXT step are issued at the end of a step, so their latency (and the K-fragment LDS reads)
// hides under the staging stores and the barrier instead of sitting in front of the VALU-bound softmax; K therefore
// runs one step ahead of V through the LDS ring.
// NW = waves (x 32 queries) per block: 4, or 8 for the estimator's launches while other chains share the chip -- one block per
// (batch, head) at T <= 256 instead of two, i.e. the same waves on HALF the CUs with two waves per SIMD (a 4-wave block per CU runs
// one wave per SIMD with every latency exposed and still owns the CU's time), and every K / V tile staged once per 256 queries.
// WHOLE (additive-bias form, L <= 64 NBUF): every K / V tile of the (batch, head) is staged ONCE in the prologue (NBUF tiles each)
// and the step loop runs without staging stores, prefetch waits or barriers -- at the estimator's mid-block length (T = 250: four
// tiles) a block's whole K and V are 74 KB, and the loop's per-step barrier + staging were a third of a step.
template <bool REL, bool DROP, int NBUF, int NW = 4, bool WHOLE = false>
DEV void attn32_fwd_body(const AP<bf16_t>& p, const int bid) {
    using namespace a32;
    static_assert(NW == 4 || (NW == 8 && !REL), "eight-wave blocks: additive-bias (estimator) form only");
    static_assert(!WHOLE || !REL, "whole-sequence staging: additive-bias (estimator) form only");
    constexpr int NT_ = 64 * NW, QB = 32 * NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
    bf16_t* Vs = Ks + NBUF * TILE;
    float* Gs = reinterpret_cast<float*>(Vs + NBUF * TILE);       // REL: 4 x 32 x SKW

    STAMP_DECL();
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const int L = p.L, nblk = (L + QB - 1) / QB;
    int bh, blk;
    block_map(bid, nblk, p.B * p.H, bh, blk);
    if (REL && p.causal) blk = nblk - 1 - blk;      // longest blocks first
    const int i0 = blk * QB, hh = bh % p.H, b = bh / p.H;
    const int iw = i0 + 32 * w, i = iw + l31;
    const size_t rowbase = (size_t)b * L;
    const bf16_t* qg = p.q + rowbase * p.ld + hh * 64;
    const bf16_t* kg = p.k + rowbase * p.ld + hh * 64;
    const bf16_t* vg = p.v + rowbase * p.ld + hh * 64;
    const bf16_t* pg = REL ? p.p + hh * 64 : nullptr;
    const int lb = p.len ? p.len[b] : L;
    const DropCtx dc = drop_ctx<DROP>(p);

    int jmax;                                   // keys the block stages
    if (REL) {
        jmax = min(L, lb);
        if (p.causal) jmax = min(jmax, i0 + QB);
    } else {
        jmax = (lb >= 1) ? min(L, lb) : L;
    }
    int jwv = jmax;                             // keys this wave's queries can see
    if (REL && p.causal) jwv = min(jwv, iw + 32);
    if (iw >= L) jwv = 0;
    const int jlim = min(L, lb);
    const int nt = (jmax + 63) >> 6;
    const float c2 = p.scale * LOG2E;

    bf16x8 qu[4], qv[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const bf16x8 f = gfrag(qg, i, L, p.ld, s, hf);
        qu[s] = f;
        qv[s] = f;
        if (REL) {
            qu[s] = bias_frag(f, p.bu + hh * 64, s, hf);
            qv[s] = bias_frag(f, p.bv + hh * 64, s, hf);
        }
    }
    const int M2 = 2 * L - 2;
    const int mlo0 = (L - 1) - iw - 31;
    f32x16 gcarry = zero16();
    bf16x8 pf0[4], pf1[4];
    Tile2 kr, vr;
    if constexpr (WHOLE) {      // all loads first (one memory round trip), then all stores, one barrier
        Tile2 ka[NBUF], va[NBUF];
#pragma unroll
        for (int t = 0; t < NBUF; ++t) {
            tload<NT_>(ka[t], kg, p.ld, 64 * t, L, tid);
            tload<NT_>(va[t], vg, p.ld, 64 * t, L, tid);
        }
#pragma unroll
        for (int t = 0; t < NBUF; ++t) {
            tstore<NT_>(ka[t], Ks + t * TILE, tid);
            tstore<NT_>(va[t], Vs + t * TILE, tid);
        }
        touch(qu);
        __syncthreads();
    } else {
    tload<NT_>(kr, kg, p.ld, 0, L, tid);      // unconditional: nothing here waits for len[b]
    tload<NT_>(vr, vg, p.ld, 0, L, tid);
    }
    if (REL) {
        bf16x8 t0[4];
        wave_frag_load(t0, pg, p.ldp, mlo0 + l31, M2 + 1, hf);
        wave_frag_load(pf0, pg, p.ldp, mlo0 + 32 + l31, M2 + 1, hf);
        wave_frag_load(pf1, pg, p.ldp, mlo0 + 64 + l31, M2 + 1, hf);
#pragma unroll
        for (int s = 0; s < 4; ++s) gcarry = mfma(t0[s], qv[s], gcarry);
    }
    if constexpr (!WHOLE) {
    tstore<NT_>(kr, Ks, tid);
    touch(qu);
    if (REL) touch(qv);
    tload<NT_>(kr, kg, p.ld, 64, L, tid);
    __syncthreads();
    }

    float* Gw = Gs + w * 32 * SKW + l31 * SKW;
    f32x16 s[2];
    s[0] = zero16();
    s[1] = zero16();
    if (0 < jwv) {
        fwd_scores<REL>(s, Ks, qu, qv, pf0, pf1, gcarry, Gw, lane);
        if (REL) {
            wave_frag_load(pf0, pg, p.ldp, mlo0 + 96 + l31, M2 + 1, hf);
            wave_frag_load(pf1, pg, p.ldp, mlo0 + 128 + l31, M2 + 1, hf);
        }
    }
    if constexpr (!WHOLE) {
    if (NBUF == 1) __syncthreads();
    tstore<NT_>(kr, Ks + (1 % NBUF) * TILE, tid);
    tstore<NT_>(vr, Vs, tid);
    tload<NT_>(kr, kg, p.ld, 128, L, tid);
    tload<NT_>(vr, vg, p.ld, 64, L, tid);
    __syncthreads();
    }

    float m_run = NINF, l_run = 0.f;
    f32x16 oacc[2], lacc;
    oacc[0] = zero16();
    oacc[1] = zero16();
    lacc = zero16();
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
    STAMP(0);      // prologue

    for (int t = 0; t < nt; ++t) {
        const int j0 = 64 * t;
        const bf16_t* Vc = Vs + (t % NBUF) * TILE;
        const bf16_t* Kn = Ks + ((t + 1) % NBUF) * TILE;
        if (j0 < jwv) {     // wave-uniform
            bool interior;
            if (REL) interior = (j0 + 63 < jlim) && (!p.causal || j0 + 63 <= iw);
            else interior = (j0 + 63 < jlim) && p.iso <= 0;
            float mx = NINF;
            if (interior) {      // raw scores: the scale rides in the exponent's FMA (scale > 0: max commutes)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
                mx *= c2;
            } else {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int j = j0 + 32 * kt + rowof(r) + 4 * hf;
                        float x = s[kt][r] * c2;
                        if (REL) {
                            const bool valid = (j < jlim) & (!p.causal | (j <= i));
                            x = valid ? x : NINF;
                        } else {
                            x = (j < L) ? x + (j < lb ? 0.f : NEGB2) : NINF;
                            if (p.iso > 0 && ((i < p.iso) != (j < p.iso))) x = NINF;
                        }
                        s[kt][r] = x;
                        mx = fmaxf(mx, x);
                    }
            }
            mx = xh_max(mx);
            const float mn = fmaxf(m_run, mx);
            const float ms = (mn == NINF) ? 0.f : mn;
            const float alpha = ex2(m_run - ms);
            m_run = mn;
            STAMP(2);      // max
            if (interior) {
                const f32x2 c2v = {c2, c2}, msv = {ms, ms};
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        f32x2 x = {s[kt][r], s[kt][r + 1]};
                        x = x * c2v - msv;      // v_pk_fma_f32
                        s[kt][r] = ex2(x[0]);
                        s[kt][r + 1] = ex2(x[1]);
                    }
            } else {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[kt][r] = ex2(s[kt][r] - ms);
            }
            STAMP(3);      // exp
            const bool resc = !__all(alpha == 1.f);
            if (resc) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
            }
            if (DROP) {     // the softmax denominator keeps the undropped sum; only the PV operand is masked
                float rs = 0.f;
                const unsigned long long ib = (((unsigned long long)b * p.H + hh) * L + i) * L;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        rs += s[kt][r];
                        const unsigned long long j = j0 + 32 * kt + rowof(r) + 4 * hf;
                        s[kt][r] *= attn_keep_scale(dc.key, ib + j, dc.thr, dc.inv);
                    }
                l_run = l_run * alpha + xh_sum(rs);
            } else if (resc) {
                lacc[0] *= alpha;      // (only register 0 is ever read: every row of lacc is the same sum)
            }
            const bf16x8 pb0 = pack8<0>(s[0]), pb1 = pack8<1>(s[0]), pb2 = pack8<0>(s[1]), pb3 = pack8<1>(s[1]);
            STAMP(4);      // softmax
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                oacc[dt] = mfma(ld_tr(Vc, 0, 32 * dt, lane), pb0, oacc[dt]);
                oacc[dt] = mfma(ld_tr(Vc, 16, 32 * dt, lane), pb1, oacc[dt]);
                oacc[dt] = mfma(ld_tr(Vc, 32, 32 * dt, lane), pb2, oacc[dt]);
                oacc[dt] = mfma(ld_tr(Vc, 48, 32 * dt, lane), pb3, oacc[dt]);
            }
            if (!DROP) {      // row sums on the matrix core: a fragment of ones against the same packed P
                lacc = mfma(ones, pb0, lacc);
                lacc = mfma(ones, pb1, lacc);
                lacc = mfma(ones, pb2, lacc);
                lacc = mfma(ones, pb3, lacc);
            }
            STAMP(5);      // PV
        }
        if (j0 + 64 < jwv) {     // wave-uniform: next step's scores
            fwd_scores<REL>(s, Kn, qu, qv, pf0, pf1, gcarry, Gw, lane);
            if (REL) {
                wave_frag_load(pf0, pg, p.ldp, mlo0 + j0 + 160 + l31, M2 + 1, hf);
                wave_frag_load(pf1, pg, p.ldp, mlo0 + j0 + 192 + l31, M2 + 1, hf);
            }
        }
        STAMP(1);          // next QK^T (+ band, skew)
        if constexpr (!WHOLE) {
        if (NBUF == 1) __syncthreads();
        STAMP(6);          // barrier 1 (single-buffer only) / idle of waves that skipped the step
        tstore<NT_>(kr, Ks + ((t + 2) % NBUF) * TILE, tid);      // (past the end: dead stores of a re-loaded last tile)
        tstore<NT_>(vr, Vs + ((t + 1) % NBUF) * TILE, tid);
        tload<NT_>(kr, kg, p.ld, 64 * (t + 3), L, tid);
        tload<NT_>(vr, vg, p.ld, 64 * (t + 2), L, tid);
        STAMP(7);          // wait for the prefetch + LDS stores + next loads issued
        __syncthreads();
        STAMP(8);          // barrier 2
        }
    }

    if (i < L) {
        const float lt = DROP ? l_run : lacc[0];
        const float inv = lt > 0.f ? 1.f / lt : 0.f;
        bf16_t* orow = p.o + (rowbase + i) * p.ldo + hh * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                st4(orow + 32 * dt + 8 * e + 4 * hf, oacc[dt][4 * e] * inv, oacc[dt][4 * e + 1] * inv,
                    oacc[dt][4 * e + 2] * inv, oacc[dt][4 * e + 3] * inv);
        if (p.o_lo) {      // what the bf16 store dropped (attn_common.h: o_lo), for the backward's delta
            bf16_t* lrow = p.o_lo + (rowbase + i) * p.ldo + hh * 64;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float r4[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float x = oacc[dt][4 * e + c] * inv;
                        r4[c] = x - (float)(bf16_t)x;
                    }
                    st4(lrow + 32 * dt + 8 * e + 4 * hf, r4[0], r4[1], r4[2], r4[3]);
                }
        }
        if (hf == 0)
            p.lse[((size_t)b * p.H + hh) * L + i] = lt > 0.f ? (m_run + __log2f(lt)) * LN2 : __builtin_inff();
    }
    STAMP(9);              // epilogue
    STAMP_FLUSH();
}
// (A grid-strided form of these launches -- a capped grid walking the owner blocks, as gemm_p256.hip does -- was measured and
// dropped: a cap of 128 / 192 / 96 resident blocks on the rel-pos launches cost the joint step +0.3 / +0.07 / +1.7 ms, and the
// loop wrapper alone slowed the rel-pos forward from 24 to 32 us per launch.)
template <bool REL, bool DROP, int NBUF, int NW = 4, bool WHOLE = false>
__global__ void __launch_bounds__(64 * NW) attn32_fwd_kernel(AP<bf16_t> p) {
    attn32_fwd_body<REL, DROP, NBUF, NW, WHOLE>(p, blockIdx.x);
}

// =====================================================================================================================
// backward dQ: block = 4 waves x 32 queries, 64 keys per step
// =====================================================================================================================
// PRE: delta arrives in p.delta (rowsum(dO . O), formed by whoever produced dO: the estimator block's tail backward) -- the O / O_lo
// rows are not read and nothing is published
template <bool REL, bool DROP, int NBUF, int NW = 4, bool WHOLE = false, bool PRE = false>
DEV void attn32_bwd_dq_body(const AP<bf16_t>& p, const int bid) {
    using namespace a32;
    static_assert(!PRE || !REL, "precomputed delta: additive-bias (estimator) form only");
    static_assert(NW == 4 || (NW == 8 && !REL), "eight-wave blocks: additive-bias (estimator) form only");
    static_assert(!WHOLE || !REL, "whole-sequence staging: additive-bias (estimator) form only");
    constexpr int NT_ = 64 * NW, QB = 32 * NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
    bf16_t* Vs = Ks + NBUF * TILE;
    bf16_t* Pb = Vs + NBUF * TILE;                                      // REL: 192 x LDK band of p rows
    float* Gs = reinterpret_cast<float*>(Pb + (REL ? 192 * LDK : 0));   // REL: 4 x 32 x SKW
    bf16_t* Rs = reinterpret_cast<bf16_t*>(Gs + (REL ? 4 * 32 * SKW : 0));   // REL: 4 x 32 x RSK

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const int L = p.L, nblk = (L + QB - 1) / QB;
    int bh, blk;
    block_map(bid, nblk, p.B * p.H, bh, blk);
    if (REL && p.causal) blk = nblk - 1 - blk;      // longest blocks first
    const int i0 = blk * QB, hh = bh % p.H, b = bh / p.H;
    const int iw = i0 + 32 * w, i = iw + l31;
    const size_t rowbase = (size_t)b * L;
    const bf16_t* qg = p.q + rowbase * p.ld + hh * 64;
    const bf16_t* kg = p.k + rowbase * p.ld + hh * 64;
    const bf16_t* vg = p.v + rowbase * p.ld + hh * 64;
    const bf16_t* dog = p.d_o + rowbase * p.ldo + hh * 64;
    const bf16_t* og = p.o + rowbase * p.ldo + hh * 64;
    const bf16_t* pg = REL ? p.p + hh * 64 : nullptr;
    const int lb = p.len ? p.len[b] : L;
    const DropCtx dc = drop_ctx<DROP>(p);

    int jmax;
    if (REL) {
        jmax = min(L, lb);
        if (p.causal) jmax = min(jmax, i0 + QB);
    } else {
        jmax = (lb >= 1) ? min(L, lb) : L;
    }
    int jwv = jmax;
    if (REL && p.causal) jwv = min(jwv, iw + 32);
    if (iw >= L) jwv = 0;
    const int jlim = min(L, lb);
    const int nt = (jmax + 63) >> 6;
    const float c2 = p.scale * LOG2E;

    bf16x8 qu[4], qv[4], dof[4];
    float dsum = 0.f;
    // (unconditional loads: without a residual buffer the O rows are read twice and the second copy weighted 0)
    const bf16_t* olg = (p.o_lo ? p.o_lo : p.o) + rowbase * p.ldo + hh * 64;
    const float lo_w = p.o_lo ? 1.f : 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const bf16x8 f = gfrag(qg, i, L, p.ld, s, hf);
        qu[s] = f;
        qv[s] = f;
        if (REL) {
            qu[s] = bias_frag(f, p.bu + hh * 64, s, hf);
            qv[s] = bias_frag(f, p.bv + hh * 64, s, hf);
        }
        dof[s] = gfrag(dog, i, L, p.ldo, s, hf);
        if (!PRE) {
            const bf16x8 of = gfrag(og, i, L, p.ldo, s, hf);
            const bf16x8 lf = gfrag(olg, i, L, p.ldo, s, hf);
#pragma unroll
            for (int e = 0; e < 8; ++e) dsum += (float)dof[s][e] * ((float)of[e] + lo_w * (float)lf[e]);
        }
    }
    if (PRE) {
        dsum = p.delta[((size_t)b * p.H + hh) * L + min(i, L - 1)];
    } else {
        // delta[i] = sum_d dO[i][d] O[i][d]; published for the dK/dV kernel, which runs after this one on the stream
        dsum = xh_sum(dsum);
        if (hf == 0 && i < L) const_cast<float*>(p.delta)[((size_t)b * p.H + hh) * L + i] = dsum;
    }
    const float lse_raw = p.lse[((size_t)b * p.H + hh) * L + min(i, L - 1)];
    const float lse2 = (i < L) ? lse_raw * LOG2E : __builtin_inff();

    float* Gw = Gs + w * 32 * SKW + l31 * SKW;
    bf16_t* Rw = Rs + w * 32 * RSK + l31 * RSK;
    if (REL) {      // the inverse-skew window is zero outside the parallelogram every step rewrites
        for (int c = hf * 4; c < RSK; c += 8) {
            bf16x4 z = {0, 0, 0, 0};
            *reinterpret_cast<bf16x4*>(Rw + c) = z;
        }
    }

    Tile2 kr, vr, br0, br1, br2;
#define A32_DQ_PREFETCH(J0_)                                                   \
    do {                                                                       \
        tload<NT_>(kr, kg, p.ld, (J0_), L, tid);                                    \
        tload<NT_>(vr, vg, p.ld, (J0_), L, tid);                                    \
        if (REL) {                                                             \
            const int mb_ = (L - 1) - (i0 + 127) + (J0_);                      \
            tload<NT_>(br0, pg, p.ldp, mb_, 2 * L - 1, tid);                        \
            tload<NT_>(br1, pg, p.ldp, mb_ + 64, 2 * L - 1, tid);                   \
            tload<NT_>(br2, pg, p.ldp, mb_ + 128, 2 * L - 1, tid);                  \
        }                                                                      \
    } while (0)
#define A32_DQ_PUBLISH(BUF_)                                                   \
    do {                                                                       \
        tstore<NT_>(kr, Ks + (BUF_) * TILE, tid);                                   \
        tstore<NT_>(vr, Vs + (BUF_) * TILE, tid);                                   \
        if (REL) {                                                             \
            tstore<NT_>(br0, Pb, tid);                                              \
            tstore<NT_>(br1, Pb + 64 * LDK, tid);                                   \
            tstore<NT_>(br2, Pb + 128 * LDK, tid);                                  \
        }                                                                      \
    } while (0)
    if constexpr (WHOLE) {      // (see attn32_fwd_body) all K / V tiles once, barrier-free step loop
        Tile2 ka[NBUF], va[NBUF];
#pragma unroll
        for (int t = 0; t < NBUF; ++t) {
            tload<NT_>(ka[t], kg, p.ld, 64 * t, L, tid);
            tload<NT_>(va[t], vg, p.ld, 64 * t, L, tid);
        }
#pragma unroll
        for (int t = 0; t < NBUF; ++t) {
            tstore<NT_>(ka[t], Ks + t * TILE, tid);
            tstore<NT_>(va[t], Vs + t * TILE, tid);
        }
        touch(qu);
        touch(dof);
        __syncthreads();
    } else {
    A32_DQ_PREFETCH(0);      // unconditional: nothing here waits for len[b]
    A32_DQ_PUBLISH(0);
    touch(qu);
    touch(dof);
    if (REL) touch(qv);
    A32_DQ_PREFETCH(64);
    __syncthreads();
    }

    f32x16 dqacc[2];
    dqacc[0] = zero16();
    dqacc[1] = zero16();
    const bf16_t* Pw = Pb + (3 - w) * 32 * LDK;      // this wave's 96 band rows

    for (int t = 0; t < nt; ++t) {
        const int j0 = 64 * t;
        const bf16_t* Kc = Ks + (t % NBUF) * TILE;
        const bf16_t* Vc = Vs + (t % NBUF) * TILE;
        if (j0 < jwv) {
            f32x16 s[2], dp[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                s[kt] = zero16();
                dp[kt] = zero16();
#pragma unroll
                for (int ss = 0; ss < 4; ++ss) {
                    s[kt] = mfma(ld_row(Kc, 32 * kt, ss, lane), qu[ss], s[kt]);
                    dp[kt] = mfma(ld_row(Vc, 32 * kt, ss, lane), dof[ss], dp[kt]);
                }
            }
            if (REL) {
#pragma unroll
                for (int mt = 0; mt < 3; ++mt) {
                    f32x16 g = zero16();
#pragma unroll
                    for (int ss = 0; ss < 4; ++ss) g = mfma(ld_row(Pw, 32 * mt, ss, lane), qv[ss], g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        f32x4 v4 = {g[4 * e], g[4 * e + 1], g[4 * e + 2], g[4 * e + 3]};
                        *reinterpret_cast<f32x4*>(Gw + 32 * mt + 8 * e + 4 * hf) = v4;
                    }
                }
                LDS_FENCE();
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[kt][r] += Gw[(31 - l31) + 32 * kt + rowof(r) + 4 * hf];
                LDS_FENCE();
            }
            bool interior;
            if (REL) interior = (j0 + 63 < jlim) && (!p.causal || j0 + 63 <= iw);
            else interior = (j0 + 63 < jlim) && p.iso <= 0;
            const unsigned long long ib = (((unsigned long long)b * p.H + hh) * L + i) * L;
            if (interior && !DROP) {      // packed form: p = exp2(s c2 - lse), dS = p (dP scale - delta scale)
                const f32x2 c2v = {c2, c2}, lv = {lse2, lse2}, scv = {p.scale, p.scale}, dsv = {dsum * p.scale, dsum * p.scale};
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        f32x2 x = {s[kt][r], s[kt][r + 1]};
                        x = x * c2v - lv;
                        const f32x2 pv = {ex2(x[0]), ex2(x[1])};
                        f32x2 d = {dp[kt][r], dp[kt][r + 1]};
                        d = d * scv - dsv;
                        d = d * pv;
                        s[kt][r] = d[0];
                        s[kt][r + 1] = d[1];
                    }
            } else {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int j = j0 + 32 * kt + rowof(r) + 4 * hf;
                        float x = s[kt][r] * c2;
                        if (!interior) {      // (wave-uniform; the per-element part is selects only, no branches)
                            if (REL) {
                                const bool valid = (j < jlim) & (!p.causal | (j <= i));
                                x = valid ? x : NINF;
                            } else {
                                const bool valid = (j < L) & !((p.iso > 0) & ((i < p.iso) != (j < p.iso)));
                                x += (j < lb ? 0.f : NEGB2);
                                x = valid ? x : NINF;
                            }
                        }
                        const float pv = ex2(x - lse2);      // masked positions: exp2(-inf) = exact 0
                        float dpe = dp[kt][r];
                        if (DROP) dpe *= attn_keep_scale(dc.key, ib + (unsigned long long)j, dc.thr, dc.inv);
                        s[kt][r] = pv * (dpe - dsum) * p.scale;      // dS
                    }
            }
            const bf16x8 db0 = pack8<0>(s[0]), db1 = pack8<1>(s[0]), db2 = pack8<0>(s[1]), db3 = pack8<1>(s[1]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dqacc[dt] = mfma(ld_tr(Kc, 0, 32 * dt, lane), db0, dqacc[dt]);
                dqacc[dt] = mfma(ld_tr(Kc, 16, 32 * dt, lane), db1, dqacc[dt]);
                dqacc[dt] = mfma(ld_tr(Kc, 32, 32 * dt, lane), db2, dqacc[dt]);
                dqacc[dt] = mfma(ld_tr(Kc, 48, 32 * dt, lane), db3, dqacc[dt]);
            }
            if (REL) {
                // dQ_i += sum_j dS[i,j] p[L-1-i+j]: dS through the inverse shift (column jl -> 31 - l31 + jl)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) Rw[(31 - l31) + 32 * kt + rowof(r) + 4 * hf] = (bf16_t)s[kt][r];
                LDS_FENCE();
#pragma unroll
                for (int kk = 0; kk < 6; ++kk) {
                    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(Rw + 16 * kk + 4 * hf);
                    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(Rw + 16 * kk + 8 + 4 * hf);
                    const bf16x8 dk8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) dqacc[dt] = mfma(ld_tr(Pw, 16 * kk, 32 * dt, lane), dk8, dqacc[dt]);
                }
                LDS_FENCE();
            }
        }
        if constexpr (!WHOLE) {
        if (NBUF == 1 || REL) __syncthreads();
        A32_DQ_PUBLISH((t + 1) % NBUF);
        A32_DQ_PREFETCH(64 * (t + 2));
        __syncthreads();
        }
    }

    if (i < L) {
        bf16_t* drow = p.dq + (rowbase + i) * p.ldg + hh * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                st4(drow + 32 * dt + 8 * e + 4 * hf, dqacc[dt][4 * e], dqacc[dt][4 * e + 1], dqacc[dt][4 * e + 2],
                    dqacc[dt][4 * e + 3]);
    }
}

template <bool REL, bool DROP, int NBUF, int NW = 4>
__global__ void __launch_bounds__(64 * NW) attn32_bwd_dq_kernel(AP<bf16_t> p) {
    attn32_bwd_dq_body<REL, DROP, NBUF, NW, false>(p, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------------
// rel-pos dQ at TWO blocks per CU (two waves per SIMD): the same arithmetic as attn32_bwd_dq_body<true, DROP, 1> in 80 896 bytes of
// LDS (was 122 880) and <= 256 registers, so that one block's LDS skew traffic, softmax and barriers overlap the other's MFMAs:
//   * the forward-skew window holds 64 of the band's 96 columns at a time: band tiles 0 and 1 serve the step's first 32 keys; tile 2
//     then overwrites tile 0's columns (ring of two 32-column halves) and serves the second 32 -- no band product is repeated;
//   * the inverse-skew window (dS, bf16) lives in the SAME per-wave bytes: the g values are dead when dS is formed; its rows are
//     cleared and rewritten every step (12 + 32 stores per lane);
//   * the band of p rows is a ring of three 64-row tiles: a step moves the band by 64 rows, so ONE new tile is staged per step
//     instead of three (two tiles of staging registers and two tile stores less).
// ---------------------------------------------------------------------------------------------------------------------
#ifndef REL2_PACKED
#define REL2_PACKED 0
#endif
#ifndef REL2_SYNC
#define REL2_SYNC 1
#endif
constexpr int SKH = 68;            // half-window row stride (dwords): 64 columns used; 67 l31 + const is conflict-free
template <bool DROP>
DEV void attn32_bwd_dq_rel2_body(const AP<bf16_t>& p, const int bid) {
    using namespace a32;
    constexpr int NT_ = 256, QB = 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
    bf16_t* Vs = Ks + TILE;
    bf16_t* Pb = Vs + TILE;                                             // ring of three 64 x LDK tiles of p rows
    float* Gs = reinterpret_cast<float*>(Pb + 192 * LDK);               // 4 x 32 x SKH (and, aliased, the dS window)

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hf = lane >> 5;
    // (scalar wave index: the ring-slot pointers and the wave's key limit stay in SGPRs -- 248 registers, no spill; the dropout
    //  instantiation allocates better with the vector form: 248 / 0 against 256 / 13 spills.  tools/isa_census.py shows both.)
    const int w = DROP ? (tid >> 6) : __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = p.L, nblk = (L + QB - 1) / QB;
    int bh, blk;
    block_map(bid, nblk, p.B * p.H, bh, blk);
    if (p.causal) blk = nblk - 1 - blk;             // longest blocks first
    const int i0 = blk * QB, hh = bh % p.H, b = bh / p.H;
    const int iw = i0 + 32 * w, i = iw + l31;
    const size_t rowbase = (size_t)b * L;
    const bf16_t* qg = p.q + rowbase * p.ld + hh * 64;
    const bf16_t* kg = p.k + rowbase * p.ld + hh * 64;
    const bf16_t* vg = p.v + rowbase * p.ld + hh * 64;
    const bf16_t* dog = p.d_o + rowbase * p.ldo + hh * 64;
    const bf16_t* og = p.o + rowbase * p.ldo + hh * 64;
    const bf16_t* pg = p.p + hh * 64;
    const int lb = p.len ? p.len[b] : L;
    const DropCtx dc = drop_ctx<DROP>(p);

    int jmax = min(L, lb);
    if (p.causal) jmax = min(jmax, i0 + QB);
    int jwv = jmax;
    if (p.causal) jwv = min(jwv, iw + 32);
    if (iw >= L) jwv = 0;
    const int jlim = min(L, lb);
    const int nt = (jmax + 63) >> 6;
    const float c2 = p.scale * LOG2E;

    bf16x8 qu[4], qv[4], dof[4];
    float dsum = 0.f;
    const bf16_t* olg = (p.o_lo ? p.o_lo : p.o) + rowbase * p.ldo + hh * 64;
    const float lo_w = p.o_lo ? 1.f : 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const bf16x8 f = gfrag(qg, i, L, p.ld, s, hf);
        qu[s] = bias_frag(f, p.bu + hh * 64, s, hf);
        qv[s] = bias_frag(f, p.bv + hh * 64, s, hf);
        dof[s] = gfrag(dog, i, L, p.ldo, s, hf);
        const bf16x8 of = gfrag(og, i, L, p.ldo, s, hf);
        const bf16x8 lf = gfrag(olg, i, L, p.ldo, s, hf);
#pragma unroll
        for (int e = 0; e < 8; ++e) dsum += (float)dof[s][e] * ((float)of[e] + lo_w * (float)lf[e]);
    }
    dsum = xh_sum(dsum);
    if (hf == 0 && i < L) const_cast<float*>(p.delta)[((size_t)b * p.H + hh) * L + i] = dsum;
    const float lse_raw = p.lse[((size_t)b * p.H + hh) * L + min(i, L - 1)];
    const float lse2 = (i < L) ? lse_raw * LOG2E : __builtin_inff();

    float* Gw = Gs + w * 32 * SKH + l31 * SKH;                          // this lane's row of the g half-window (fp32)
    bf16_t* Rw = reinterpret_cast<bf16_t*>(Gs + w * 32 * SKH) + l31 * RSK;   // ... and of the dS window (bf16), same bytes

    const int mb0 = (L - 1) - (i0 + 127);          // band row of logical tile 0 at step 0
    Tile2 kr, vr, br;
    {
        Tile2 b0, b1;
        tload<NT_>(kr, kg, p.ld, 0, L, tid);
        tload<NT_>(vr, vg, p.ld, 0, L, tid);
        tload<NT_>(b0, pg, p.ldp, mb0, 2 * L - 1, tid);
        tload<NT_>(b1, pg, p.ldp, mb0 + 64, 2 * L - 1, tid);
        tload<NT_>(br, pg, p.ldp, mb0 + 128, 2 * L - 1, tid);
        tstore<NT_>(kr, Ks, tid);
        tstore<NT_>(vr, Vs, tid);
        tstore<NT_>(b0, Pb, tid);
        tstore<NT_>(b1, Pb + 64 * LDK, tid);
        tstore<NT_>(br, Pb + 128 * LDK, tid);
    }
    touch(qu);
    touch(dof);
    touch(qv);
    if (!REL2_SYNC) {
        tload<NT_>(kr, kg, p.ld, 64, L, tid);
        tload<NT_>(vr, vg, p.ld, 64, L, tid);
        tload<NT_>(br, pg, p.ldp, mb0 + 192, 2 * L - 1, tid);
    }
    __syncthreads();

    f32x16 dqacc[2];
    dqacc[0] = zero16();
    dqacc[1] = zero16();
    int t3 = 0;                                     // t % 3: ring slot of this step's logical band tile 0

    for (int t = 0; t < nt; ++t) {
        const int j0 = 64 * t;
        if (j0 < jwv) {
            // this wave's 96 band rows = 32-row chunks 3 - w + {0, 1, 2} of the six the ring holds; chunk c sits in slot (t3 + c / 2) % 3
            const bf16_t* Pc[3];
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
                const int c = 3 - w + mt;
                int slot = t3 + (c >> 1);
                slot = slot >= 3 ? slot - 3 : slot;
                Pc[mt] = Pb + (slot * 64 + (c & 1) * 32) * LDK;
            }
            f32x16 s[2], dp[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                s[kt] = zero16();
                dp[kt] = zero16();
#pragma unroll
                for (int ss = 0; ss < 4; ++ss) {
                    s[kt] = mfma(ld_row(Ks, 32 * kt, ss, lane), qu[ss], s[kt]);
                    dp[kt] = mfma(ld_row(Vs, 32 * kt, ss, lane), dof[ss], dp[kt]);
                }
            }
            // band tiles 0, 1 -> window halves 0, 1 -> keys 0..31; band tile 2 -> half 0 -> keys 32..63 (columns taken modulo 64)
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
                f32x16 g = zero16();
#pragma unroll
                for (int ss = 0; ss < 4; ++ss) g = mfma(ld_row(Pc[mt], 0, ss, lane), qv[ss], g);
                if (mt == 2) {
                    LDS_FENCE();
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[0][r] += Gw[(31 - l31) + rowof(r) + 4 * hf];
                    LDS_FENCE();
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f32x4 v4 = {g[4 * e], g[4 * e + 1], g[4 * e + 2], g[4 * e + 3]};
                    *reinterpret_cast<f32x4*>(Gw + 32 * (mt & 1) + 8 * e + 4 * hf) = v4;
                }
            }
            LDS_FENCE();
#pragma unroll
            for (int r = 0; r < 16; ++r) s[1][r] += Gw[((31 - l31) + 32 + rowof(r) + 4 * hf) & 63];
            LDS_FENCE();
            const bool interior = (j0 + 63 < jlim) && (!p.causal || j0 + 63 <= iw);
            const unsigned long long ib = (((unsigned long long)b * p.H + hh) * L + i) * L;
            if (REL2_PACKED && interior && !DROP) {      // packed form: p = exp2(s c2 - lse), dS = p (dP scale - delta scale)
                const f32x2 c2v = {c2, c2}, lv = {lse2, lse2}, scv = {p.scale, p.scale}, dsv = {dsum * p.scale, dsum * p.scale};
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        f32x2 x = {s[kt][r], s[kt][r + 1]};
                        x = x * c2v - lv;
                        const f32x2 pv = {ex2(x[0]), ex2(x[1])};
                        f32x2 d = {dp[kt][r], dp[kt][r + 1]};
                        d = d * scv - dsv;
                        d = d * pv;
                        s[kt][r] = d[0];
                        s[kt][r + 1] = d[1];
                    }
            } else {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int j = j0 + 32 * kt + rowof(r) + 4 * hf;
                        float x = s[kt][r] * c2;
                        if (!interior) {      // (wave-uniform; the per-element part is selects only, no branches)
                            const bool valid = (j < jlim) & (!p.causal | (j <= i));
                            x = valid ? x : NINF;
                        }
                        const float pv = ex2(x - lse2);      // masked positions: exp2(-inf) = exact 0
                        float dpe = dp[kt][r];
                        if (DROP) dpe *= attn_keep_scale(dc.key, ib + (unsigned long long)j, dc.thr, dc.inv);
                        s[kt][r] = pv * (dpe - dsum) * p.scale;      // dS
                    }
            }
            const bf16x8 db0 = pack8<0>(s[0]), db1 = pack8<1>(s[0]), db2 = pack8<0>(s[1]), db3 = pack8<1>(s[1]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dqacc[dt] = mfma(ld_tr(Ks, 0, 32 * dt, lane), db0, dqacc[dt]);
                dqacc[dt] = mfma(ld_tr(Ks, 16, 32 * dt, lane), db1, dqacc[dt]);
                dqacc[dt] = mfma(ld_tr(Ks, 32, 32 * dt, lane), db2, dqacc[dt]);
                dqacc[dt] = mfma(ld_tr(Ks, 48, 32 * dt, lane), db3, dqacc[dt]);
            }
            // dQ_i += sum_j dS[i,j] p[L-1-i+j]: dS through the inverse shift (column jl -> 31 - l31 + jl); the row is cleared first
            // (the g values sat in these bytes), 48 columns per lane half
#pragma unroll
            for (int c = 0; c < 12; ++c) {
                const bf16x4 z = {0, 0, 0, 0};
                *reinterpret_cast<bf16x4*>(Rw + 48 * hf + 4 * c) = z;
            }
            LDS_FENCE();
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) Rw[(31 - l31) + 32 * kt + rowof(r) + 4 * hf] = (bf16_t)s[kt][r];
            LDS_FENCE();
#pragma unroll
            for (int kk = 0; kk < 6; ++kk) {
                const bf16x4 lo = *reinterpret_cast<const bf16x4*>(Rw + 16 * kk + 4 * hf);
                const bf16x4 hi = *reinterpret_cast<const bf16x4*>(Rw + 16 * kk + 8 + 4 * hf);
                const bf16x8 dk8 = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) dqacc[dt] = mfma(ld_tr(Pc[kk >> 1], 16 * (kk & 1), 32 * dt, lane), dk8, dqacc[dt]);
            }
            LDS_FENCE();
        }
        __syncthreads();
        // publish step t + 1: K / V, and the ONE new band tile into the slot this step's logical tile 0 leaves.
        // REL2_SYNC: the tiles are requested only now (no staging registers live across the step: the block fits 256 registers
        // without spills); the wait is the other resident block's time
        if (REL2_SYNC) {
            if (t + 1 < nt) {
                tload<NT_>(kr, kg, p.ld, 64 * (t + 1), L, tid);
                tload<NT_>(vr, vg, p.ld, 64 * (t + 1), L, tid);
                tload<NT_>(br, pg, p.ldp, mb0 + 64 * (t + 1) + 128, 2 * L - 1, tid);
                tstore<NT_>(kr, Ks, tid);
                tstore<NT_>(vr, Vs, tid);
                tstore<NT_>(br, Pb + t3 * 64 * LDK, tid);
            }
            t3 = t3 == 2 ? 0 : t3 + 1;
        } else {
            tstore<NT_>(kr, Ks, tid);
            tstore<NT_>(vr, Vs, tid);
            tstore<NT_>(br, Pb + t3 * 64 * LDK, tid);
            t3 = t3 == 2 ? 0 : t3 + 1;
            tload<NT_>(kr, kg, p.ld, 64 * (t + 2), L, tid);
            tload<NT_>(vr, vg, p.ld, 64 * (t + 2), L, tid);
            tload<NT_>(br, pg, p.ldp, mb0 + 64 * (t + 2) + 128, 2 * L - 1, tid);
        }
        __syncthreads();
    }

    if (i < L) {
        bf16_t* drow = p.dq + (rowbase + i) * p.ldg + hh * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                st4(drow + 32 * dt + 8 * e + 4 * hf, dqacc[dt][4 * e], dqacc[dt][4 * e + 1], dqacc[dt][4 * e + 2],
                    dqacc[dt][4 * e + 3]);
    }
}
template <bool DROP>
__global__ void __launch_bounds__(256, 2) attn32_bwd_dq_rel2_kernel(AP<bf16_t> p) {
    attn32_bwd_dq_rel2_body<DROP>(p, blockIdx.x);
}

// =====================================================================================================================
// backward dK, dV: block = 4 waves x 32 keys, 64 queries per step.  OWN_DELTA: delta = rowsum(dO * O) of each staged
// query tile is formed here from the prefetch registers instead of read from the dQ kernel's output -- the two backward
// roles then have no dependency and can share one launch (attn32_bwd_fused_kernel).
// =====================================================================================================================
// LEAN (attn32_bwd_dkv_rel2_kernel: two blocks per CU, 256 registers): no staging registers live across a step (the next tiles are
// requested behind the step's barrier: the wait is the other resident block's time) and no packed-softmax twin of the element loop
template <bool REL, bool DROP, int NBUF, bool OWN_DELTA, int NW = 4, bool WHOLE = false, bool LEAN = false>
DEV void attn32_bwd_dkv_body(const AP<bf16_t>& p, const int bid) {
    using namespace a32;
    static_assert(NW == 4 || (NW == 8 && !REL), "eight-wave blocks: additive-bias (estimator) form only");
    static_assert(!WHOLE || !REL, "whole-sequence staging: additive-bias (estimator) form only");
    constexpr int NT_ = 64 * NW, QB = 32 * NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Qs = reinterpret_cast<bf16_t*>(smem);
    bf16_t* Os = Qs + NBUF * TILE;
    float* lse_s = reinterpret_cast<float*>(Os + NBUF * TILE);          // NBUF x 64, log2 units
    float* del_s = lse_s + NBUF * 64;                                   // NBUF x 64, delta * scale
    float* Gs = del_s + NBUF * 64;                                      // REL: 4 x 32 x SK2

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const int L = p.L, nblk = (L + QB - 1) / QB;
    int bh, blk;
    block_map(bid, nblk, p.B * p.H, bh, blk);      // (causal: key block 0 is the longest and already first)
    const int j0b = blk * QB, hh = bh % p.H, b = bh / p.H;
    const int jw = j0b + 32 * w, j = jw + l31;
    const size_t rowbase = (size_t)b * L;
    const bf16_t* qg = p.q + rowbase * p.ld + hh * 64;
    const bf16_t* kg = p.k + rowbase * p.ld + hh * 64;
    const bf16_t* vg = p.v + rowbase * p.ld + hh * 64;
    const bf16_t* dog = p.d_o + rowbase * p.ldo + hh * 64;
    const bf16_t* pg = REL ? p.p + hh * 64 : nullptr;
    const int lb = p.len ? p.len[b] : L;
    const DropCtx dc = drop_ctx<DROP>(p);
    const int lb_eff = REL ? min(L, lb) : ((lb >= 1) ? min(L, lb) : L);
    const float c2 = p.scale * LOG2E;

    f32x16 dkacc[2], dvacc[2];
    dkacc[0] = zero16(); dkacc[1] = zero16(); dvacc[0] = zero16(); dvacc[1] = zero16();
    float csum = 0.f;

    if (j0b < lb_eff) {      // block-uniform
        const bool wact = jw < lb_eff;      // wave-uniform: keys at or past the length contribute exact zeros
        bf16x8 kf[4], vf[4], ub[4], vb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[s] = gfrag(kg, j, L, p.ld, s, hf);
            vf[s] = gfrag(vg, j, L, p.ld, s, hf);
            if (REL) {
                ub[s] = const_frag(p.bu + hh * 64, s, hf);
                vb[s] = const_frag(p.bv + hh * 64, s, hf);
            }
        }
        f32x16 sinit = zero16();      // every row = u . k_j
        if (REL) {
#pragma unroll
            for (int s = 0; s < 4; ++s) sinit = mfma(ub[s], kf[s], sinit);
        }
        const float kbias = (!REL && j >= lb) ? NEGB2 : 0.f;
        const int ibeg = (REL && p.causal) ? j0b : 0;
        const int nt = (L - ibeg + 63) >> 6;

        // band fragments (B operand rows of p): query step n (32 queries from iq = ibeg + 32n) needs rows
        // mlo(n) .. mlo(n)+63, mlo(n) = L-1 - iq - 31 + jw; the upper 32 are the previous step's lower 32.
        const int M2 = 2 * L - 2;
        auto ptile = [&](bf16x8 (&f)[4], int m0) __attribute__((always_inline)) {
            const int row = min(max(m0 + l31, 0), M2);
#pragma unroll
            for (int s = 0; s < 4; ++s) f[s] = *reinterpret_cast<const bf16x8*>(pg + (size_t)row * p.ldp + 16 * s + 8 * hf);
        };
        bf16x8 pa[4], pb_[4], pn[4];
        if (REL) {
            const int mlo = (L - 1) - ibeg - 31 + jw;
            ptile(pa, mlo);
            ptile(pb_, mlo + 32);
        }

        Tile2 qr, dor, orr, olr;
        float lse_n = 0.f, del_n = 0.f;
        const bf16_t* og = p.o + rowbase * p.ldo + hh * 64;
        const bf16_t* olg = (p.o_lo ? p.o_lo : p.o) + rowbase * p.ldo + hh * 64;      // (see the dQ body: second copy weighted 0 when off)
        const float lo_w = p.o_lo ? 1.f : 0.f;
#define A32_KV_PREFETCH(I0_)                                                               \
    do {                                                                                   \
        tload<NT_>(qr, qg, p.ld, (I0_), L, tid);                                                \
        tload<NT_>(dor, dog, p.ldo, (I0_), L, tid);                                             \
        const int ii_ = min((I0_) + (tid & 63), L - 1);      /* every thread loads */      \
        lse_n = p.lse[((size_t)b * p.H + hh) * L + ii_];                                   \
        if (OWN_DELTA) { tload<NT_>(orr, og, p.ldo, (I0_), L, tid); tload<NT_>(olr, olg, p.ldo, (I0_), L, tid); } \
        else del_n = p.delta[((size_t)b * p.H + hh) * L + ii_];                            \
    } while (0)
#define A32_KV_PUBLISH(BUF_, I0_)                                                          \
    do {                                                                                   \
        tstore<NT_>(qr, Qs + (BUF_) * TILE, tid);                                               \
        tstore<NT_>(dor, Os + (BUF_) * TILE, tid);                                              \
        if (tid < 64) {                                                                    \
            const bool ok_ = (I0_) + tid < L;                                              \
            lse_s[(BUF_) * 64 + tid] = ok_ ? lse_n * LOG2E : __builtin_inff();             \
            if (!OWN_DELTA) del_s[(BUF_) * 64 + tid] = ok_ ? del_n * p.scale : 0.f;        \
        }                                                                                  \
        if (OWN_DELTA) {      /* thread: 8 columns of row tid/8 (and, 256 threads, tid/8 + 32) */ \
            _Pragma("unroll") for (int i_ = 0; i_ < 512 / NT_; ++i_) {                     \
                const bf16x8 d8_ = __builtin_bit_cast(bf16x8, dor[i_]), o8_ = __builtin_bit_cast(bf16x8, orr[i_]); \
                const bf16x8 l8_ = __builtin_bit_cast(bf16x8, olr[i_]);                    \
                float s_ = 0.f;                                                            \
                _Pragma("unroll") for (int e_ = 0; e_ < 8; ++e_) s_ += (float)d8_[e_] * ((float)o8_[e_] + lo_w * (float)l8_[e_]); \
                s_ += __shfl_xor(s_, 1);                                                   \
                s_ += __shfl_xor(s_, 2);                                                   \
                s_ += __shfl_xor(s_, 4);                                                   \
                const int row_ = (tid >> 3) + (NT_ / 8) * i_;                              \
                if ((tid & 7) == 0) del_s[(BUF_) * 64 + row_] = ((I0_) + row_ < L) ? s_ * p.scale : 0.f; \
            }                                                                              \
        }                                                                                  \
    } while (0)
        if constexpr (WHOLE) {      // (see attn32_fwd_body) all Q / dO tiles (+ lse, delta) once, barrier-free step loop
            Tile2 qa4[NBUF], da4[NBUF], oa4[OWN_DELTA ? NBUF : 1], la4[OWN_DELTA ? NBUF : 1];
            float ls4[NBUF], dl4[NBUF];
#pragma unroll
            for (int t = 0; t < NBUF; ++t) {
                tload<NT_>(qa4[t], qg, p.ld, 64 * t, L, tid);
                tload<NT_>(da4[t], dog, p.ldo, 64 * t, L, tid);
                if (OWN_DELTA) {
                    tload<NT_>(oa4[t], og, p.ldo, 64 * t, L, tid);
                    tload<NT_>(la4[t], olg, p.ldo, 64 * t, L, tid);
                }
                const size_t li = ((size_t)b * p.H + hh) * L + min(64 * t + (tid & 63), L - 1);
                ls4[t] = p.lse[li];
                dl4[t] = OWN_DELTA ? 0.f : p.delta[li];
            }
#pragma unroll
            for (int t = 0; t < NBUF; ++t) {
                qr[0] = qa4[t][0]; qr[1] = qa4[t][1];
                dor[0] = da4[t][0]; dor[1] = da4[t][1];
                if (OWN_DELTA) {
                    orr[0] = oa4[t][0]; orr[1] = oa4[t][1];
                    olr[0] = la4[t][0]; olr[1] = la4[t][1];
                }
                lse_n = ls4[t];
                del_n = dl4[t];
                A32_KV_PUBLISH(t, 64 * t);
            }
            touch(kf);
            touch(vf);
            __syncthreads();
        } else {
        A32_KV_PREFETCH(ibeg);
        A32_KV_PUBLISH(0, ibeg);
        touch(kf);
        touch(vf);
        if (REL) {
            touch(vb);
            touch(pa);
            touch(pb_);
        }
        if (!LEAN) A32_KV_PREFETCH(ibeg + 64);
        __syncthreads();
        }
        float* Gw = Gs + w * 32 * SK2;

        for (int t = 0; t < nt; ++t) {
            const int i0 = ibeg + 64 * t;
            const bf16_t* Qc = Qs + (t % NBUF) * TILE;
            const bf16_t* Oc = Os + (t % NBUF) * TILE;
            const float* lsc = lse_s + (t % NBUF) * 64;
            const float* dlc = del_s + (t % NBUF) * 64;
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const int iq = i0 + 32 * qt;
                if (REL) ptile(pn, (L - 1) - (iq + 32) - 31 + jw);      // next query step's new rows
                const bool act = wact && iq < L && !(REL && p.causal && iq + 31 < jw);     // wave-uniform
                if (act) {
                    bf16x8 qa[4];
#pragma unroll
                    for (int ss = 0; ss < 4; ++ss) qa[ss] = ld_row(Qc, 32 * qt, ss, lane);
                    f32x16 s = REL ? sinit : zero16();
                    f32x16 dp = zero16();
#pragma unroll
                    for (int ss = 0; ss < 4; ++ss) {
                        s = mfma(qa[ss], kf[ss], s);
                        dp = mfma(ld_row(Oc, 32 * qt, ss, lane), vf[ss], dp);
                    }
                    if (REL) {
                        f32x16 g0 = zero16(), g1 = zero16();
#pragma unroll
                        for (int ss = 0; ss < 4; ++ss) {      // every row = v . p_m
                            g0 = mfma(vb[ss], pa[ss], g0);
                            g1 = mfma(vb[ss], pb_[ss], g1);
                        }
#pragma unroll
                        for (int ss = 0; ss < 4; ++ss) {
                            g0 = mfma(qa[ss], pa[ss], g0);
                            g1 = mfma(qa[ss], pb_[ss], g1);
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            Gw[(rowof(r) + 4 * hf) * SK2 + l31] = g0[r];
                            Gw[(rowof(r) + 4 * hf) * SK2 + 32 + l31] = g1[r];
                        }
                        LDS_FENCE();
#pragma unroll
                        for (int r = 0; r < 16; ++r) s[r] += Gw[(rowof(r) + 4 * hf) * (SK2 - 1) + 31 + l31];
                        LDS_FENCE();
                    }
                    const bool interior = (jw + 31 < lb_eff) && (jw + 31 < L) && (iq + 31 < L) &&
                                          (!(REL && p.causal) || jw + 31 <= iq) && (REL || p.iso <= 0) && (REL || lb >= 1);
                    f32x4 lq[4], dq4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        lq[e] = *reinterpret_cast<const f32x4*>(lsc + 32 * qt + 8 * e + 4 * hf);
                        dq4[e] = *reinterpret_cast<const f32x4*>(dlc + 32 * qt + 8 * e + 4 * hf);
                    }
                    if (interior && !DROP && !LEAN) {      // packed form (del_s holds delta * scale)
                        const f32x2 c2v = {c2, c2}, scv = {p.scale, p.scale};
                        f32x2 cs2 = {0.f, 0.f};
#pragma unroll
                        for (int r = 0; r < 16; r += 2) {
                            const f32x2 lv = {lq[r >> 2][r & 3], lq[r >> 2][(r & 3) + 1]};
                            const f32x2 dv = {dq4[r >> 2][r & 3], dq4[r >> 2][(r & 3) + 1]};
                            f32x2 x = {s[r], s[r + 1]};
                            x = x * c2v - lv;
                            const f32x2 pv = {ex2(x[0]), ex2(x[1])};
                            f32x2 d = {dp[r], dp[r + 1]};
                            d = d * scv - dv;
                            d = d * pv;
                            cs2 += d;
                            s[r] = pv[0];
                            s[r + 1] = pv[1];
                            dp[r] = d[0];
                            dp[r + 1] = d[1];
                        }
                        csum += cs2[0] + cs2[1];
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int ii = iq + rowof(r) + 4 * hf;
                            float x = s[r] * c2;
                            if (!interior) {
                                if (REL) {
                                    const bool valid = (j < lb_eff) & (ii < L) & (!p.causal | (j <= ii));
                                    x = valid ? x : NINF;
                                } else {
                                    const bool valid = (j < L) & (ii < L) & !((p.iso > 0) & ((ii < p.iso) != (j < p.iso)));
                                    x += kbias;
                                    x = valid ? x : NINF;
                                }
                            }
                            const float pv = ex2(x - lq[r >> 2][r & 3]);      // rows past L carry lse = +inf
                            float ksc = 1.f;
                            if (DROP)
                                ksc = attn_keep_scale(dc.key, (((unsigned long long)b * p.H + hh) * L + ii) * L + j, dc.thr, dc.inv);
                            const float ds = pv * (dp[r] * ksc * p.scale - dq4[r >> 2][r & 3]);
                            s[r] = pv * ksc;
                            dp[r] = ds;
                            csum += ds;
                        }
                    }
                    const bf16x8 pb0 = pack8<0>(s), pb1 = pack8<1>(s), db0 = pack8<0>(dp), db1 = pack8<1>(dp);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        dvacc[dt] = mfma(ld_tr(Oc, 32 * qt, 32 * dt, lane), pb0, dvacc[dt]);
                        dvacc[dt] = mfma(ld_tr(Oc, 32 * qt + 16, 32 * dt, lane), pb1, dvacc[dt]);
                        dkacc[dt] = mfma(ld_tr(Qc, 32 * qt, 32 * dt, lane), db0, dkacc[dt]);
                        dkacc[dt] = mfma(ld_tr(Qc, 32 * qt + 16, 32 * dt, lane), db1, dkacc[dt]);
                    }
                }
                if (REL) {
#pragma unroll
                    for (int ss = 0; ss < 4; ++ss) {
                        pb_[ss] = pa[ss];
                        pa[ss] = pn[ss];
                    }
                }
            }
            if constexpr (!WHOLE) {
            if (NBUF == 1) __syncthreads();
            if (LEAN) {
                if (t + 1 < nt) {
                    A32_KV_PREFETCH(ibeg + 64 * (t + 1));
                    A32_KV_PUBLISH((t + 1) % NBUF, ibeg + 64 * (t + 1));
                }
            } else {
                A32_KV_PUBLISH((t + 1) % NBUF, ibeg + 64 * (t + 1));
                A32_KV_PREFETCH(ibeg + 64 * (t + 2));
            }
            __syncthreads();
            }
        }
        if (REL) {      // dK_j += (sum_i dS[i,j]) u   (the A operand of dK was raw q)
            const float cs = xh_sum(csum);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) dkacc[dt][r] += cs * p.bu[hh * 64 + 32 * dt + rowof(r) + 4 * hf];
        }
    }
    if (j < L) {
        bf16_t* krow = p.dk + (rowbase + j) * p.ldg + hh * 64;
        bf16_t* vrow = p.dv + (rowbase + j) * p.ldg + hh * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                st4(krow + 32 * dt + 8 * e + 4 * hf, dkacc[dt][4 * e], dkacc[dt][4 * e + 1], dkacc[dt][4 * e + 2],
                    dkacc[dt][4 * e + 3]);
                st4(vrow + 32 * dt + 8 * e + 4 * hf, dvacc[dt][4 * e], dvacc[dt][4 * e + 1], dvacc[dt][4 * e + 2],
                    dvacc[dt][4 * e + 3]);
            }
    }
}

template <bool REL, bool DROP, int NBUF>
__global__ void __launch_bounds__(256) attn32_bwd_dkv_kernel(AP<bf16_t> p) {
    attn32_bwd_dkv_body<REL, DROP, NBUF, false>(p, blockIdx.x);
}
// the rel-pos dK/dV role at two blocks per CU: the LEAN form of the body, 256-register budget
template <bool DROP, int NBUF>
__global__ void __launch_bounds__(256, 2) attn32_bwd_dkv_rel2_kernel(AP<bf16_t> p) {
    attn32_bwd_dkv_body<true, DROP, NBUF, false, 4, false, true>(p, blockIdx.x);
}

// Both backward roles in ONE launch (blocks [0, nq): dQ, blocks [nq, 2 nq): dK/dV with its own delta): at the estimator's
// sizes a role alone is one 4-wave block per CU (T = 250: 256 blocks), i.e. one wave per SIMD with every latency
// exposed; together they put two waves on a SIMD, and one launch boundary per attention backward disappears.
// PRE: delta is an INPUT (p.delta; p.o == nullptr): neither role reads O
template <bool REL, bool DROP, int NBUF, int NW = 4, bool WHOLE = false, bool PRE = false>
__global__ void __launch_bounds__(64 * NW) attn32_bwd_fused_kernel(AP<bf16_t> p, int nq) {
    if ((int)blockIdx.x < nq) attn32_bwd_dq_body<REL, DROP, NBUF, NW, WHOLE, PRE>(p, blockIdx.x);
    else attn32_bwd_dkv_body<REL, DROP, NBUF, !PRE, NW, WHOLE>(p, blockIdx.x - nq);
}

// =====================================================================================================================
// host side
// =====================================================================================================================
namespace a32 {
static size_t smem_fwd(bool rel, int nbuf) { return (size_t)nbuf * 2 * TILE * 2 + (rel ? 4 * 32 * SKW * 4 : 0); }
static size_t smem_dq(bool rel, int nbuf) {
    return (size_t)nbuf * 2 * TILE * 2 + (rel ? 192 * LDK * 2 + 4 * 32 * SKW * 4 + 4 * 32 * RSK * 2 : 0);
}
static size_t smem_dq_rel2() { return (size_t)2 * TILE * 2 + 192 * LDK * 2 + 4 * 32 * SKH * 4; }      // 80 896 bytes: two blocks per CU
static size_t smem_dkv(bool rel, int nbuf) { return (size_t)nbuf * 2 * TILE * 2 + nbuf * 2 * 64 * 4 + (rel ? 4 * 32 * SK2 * 4 : 0); }

template <typename K>
static int set_smem(K kernel, size_t bytes, const char* name) {
    static K seen[16];
    static int nseen = 0;
    for (int i = 0; i < nseen; ++i)
        if (seen[i] == kernel) return 0;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        cvft_set_error("%s: hipFuncSetAttribute(%zu bytes) failed: %s", name, bytes, hipGetErrorString(e));
        return -2;
    }
    if (nseen < 16) seen[nseen++] = kernel;
    return 0;
}
template <typename K>
static int launch(K kernel, size_t sm, const AP<bf16_t>& p, hipStream_t st, const char* name, int nw = 4) {
    if (set_smem(kernel, sm, name)) return -2;
    dim3 grid((unsigned)((p.L + 32 * nw - 1) / (32 * nw)) * p.H * p.B);
    hipLaunchKernelGGL(kernel, grid, dim3(64 * nw), sm, st, p);
    CVFT_LAUNCH_CHECK(name);
    return 0;
}
// whole-sequence staging for L <= 256 (CVFT_ATTN_WHOLE = 0 | 1, default 1)
static int est_whole() {
    static const int env = getenv("CVFT_ATTN_WHOLE") ? atoi(getenv("CVFT_ATTN_WHOLE")) : 1;
    return env;
}
static int est_waves() {
    static const int env = getenv("CVFT_ATTN_NW") ? atoi(getenv("CVFT_ATTN_NW")) : 0;
    return env == 8 ? 8 : 4;
}
}  // namespace a32

// rel = 0: additive key-bias attention (estimator); rel = 1: rel-pos attention.  Called from attention.hip for bf16.
int cvft_attn32_fwd(const AP<bf16_t>& p, int rel, hipStream_t st) {
    using namespace a32;
    if (!rel) {
        if (est_waves() == 8) return launch(attn32_fwd_kernel<false, false, 2, 8>, smem_fwd(false, 2), p, st, "attn32_fwd", 8);
        if (est_whole() && p.L <= 256) return launch(attn32_fwd_kernel<false, false, 4, 4, true>, smem_fwd(false, 4), p, st, "attn32_fwd_whole");
        return launch(attn32_fwd_kernel<false, false, 2>, smem_fwd(false, 2), p, st, "attn32_fwd");
    }
    if (p.drop_p > 0.f) return launch(attn32_fwd_kernel<true, true, 1>, smem_fwd(true, 1), p, st, "attn32_fwd_rel_drop");
    return launch(attn32_fwd_kernel<true, false, 1>, smem_fwd(true, 1), p, st, "attn32_fwd_rel");
}
int cvft_attn32_bwd(const AP<bf16_t>& p, int rel, hipStream_t st) {
    using namespace a32;
    int rc;
    if (!rel) {
        static int split = -1;
        if (split < 0) {
            const char* e = getenv("CVFT_ATTN_BWD_SPLIT");
            split = (e && e[0] == '1') ? 1 : 0;
        }
        if (p.o == nullptr) {      // delta precomputed by the producer of dO (include/cvft.h): the fused launch, neither role reads O
            if (split || est_waves() == 8) {
                cvft_set_error("cvft_attn_bias_bwd: o == NULL (delta given) is not available with CVFT_ATTN_BWD_SPLIT=1 / CVFT_ATTN_NW=8");
                return -1;
            }
            const int nq = ((p.L + 127) / 128) * p.H * p.B;
            if (est_whole() && p.L <= 256) {
                const size_t sm = smem_dq(false, 4) > smem_dkv(false, 4) ? smem_dq(false, 4) : smem_dkv(false, 4);
                if (set_smem(attn32_bwd_fused_kernel<false, false, 4, 4, true, true>, sm, "attn32_bwd_fused_whole_pre")) return -2;
                hipLaunchKernelGGL((attn32_bwd_fused_kernel<false, false, 4, 4, true, true>), dim3(2u * nq), dim3(256), sm, st, p, nq);
                CVFT_LAUNCH_CHECK("attn32_bwd_fused_whole_pre");
                return 0;
            }
            const size_t sm = smem_dq(false, 2) > smem_dkv(false, 2) ? smem_dq(false, 2) : smem_dkv(false, 2);
            if (set_smem(attn32_bwd_fused_kernel<false, false, 2, 4, false, true>, sm, "attn32_bwd_fused_pre")) return -2;
            hipLaunchKernelGGL((attn32_bwd_fused_kernel<false, false, 2, 4, false, true>), dim3(2u * nq), dim3(256), sm, st, p, nq);
            CVFT_LAUNCH_CHECK("attn32_bwd_fused_pre");
            return 0;
        }
        if (!split && est_waves() == 8) {
            const size_t sm = smem_dq(false, 2) > smem_dkv(false, 2) ? smem_dq(false, 2) : smem_dkv(false, 2);
            if (set_smem(attn32_bwd_fused_kernel<false, false, 2, 8>, sm, "attn32_bwd_fused")) return -2;
            const int nq = ((p.L + 255) / 256) * p.H * p.B;
            hipLaunchKernelGGL((attn32_bwd_fused_kernel<false, false, 2, 8>), dim3(2u * nq), dim3(512), sm, st, p, nq);
            CVFT_LAUNCH_CHECK("attn32_bwd_fused");
            return 0;
        }
        if (!split && est_whole() && p.L <= 256) {
            const size_t sm = smem_dq(false, 4) > smem_dkv(false, 4) ? smem_dq(false, 4) : smem_dkv(false, 4);
            if (set_smem(attn32_bwd_fused_kernel<false, false, 4, 4, true>, sm, "attn32_bwd_fused_whole")) return -2;
            const int nq = ((p.L + 127) / 128) * p.H * p.B;
            hipLaunchKernelGGL((attn32_bwd_fused_kernel<false, false, 4, 4, true>), dim3(2u * nq), dim3(256), sm, st, p, nq);
            CVFT_LAUNCH_CHECK("attn32_bwd_fused_whole");
            return 0;
        }
        if (!split) {
            const size_t sm = smem_dq(false, 2) > smem_dkv(false, 2) ? smem_dq(false, 2) : smem_dkv(false, 2);
            if (set_smem(attn32_bwd_fused_kernel<false, false, 2>, sm, "attn32_bwd_fused")) return -2;
            const int nq = ((p.L + 127) / 128) * p.H * p.B;
            hipLaunchKernelGGL((attn32_bwd_fused_kernel<false, false, 2>), dim3(2u * nq), dim3(256), sm, st, p, nq);
            CVFT_LAUNCH_CHECK("attn32_bwd_fused");
            return 0;
        }
        rc = launch(attn32_bwd_dq_kernel<false, false, 2>, smem_dq(false, 2), p, st, "attn32_bwd_dq");
        if (rc) return rc;
        return launch(attn32_bwd_dkv_kernel<false, false, 2>, smem_dkv(false, 2), p, st, "attn32_bwd_dkv");
    }
    // the dQ role at two blocks per CU (attn32_bwd_dq_rel2_body); CVFT_ATTN_DQ2=0: the one-block-per-CU form
    static const int dq2 = getenv("CVFT_ATTN_DQ2") ? atoi(getenv("CVFT_ATTN_DQ2")) : 1;
    static const int dkv2 = getenv("CVFT_ATTN_DKV2") ? atoi(getenv("CVFT_ATTN_DKV2")) : 0;
    if (p.drop_p > 0.f) {
        rc = dq2 ? launch(attn32_bwd_dq_rel2_kernel<true>, smem_dq_rel2(), p, st, "attn32_bwd_dq_rel2_drop")
                 : launch(attn32_bwd_dq_kernel<true, true, 1>, smem_dq(true, 1), p, st, "attn32_bwd_dq_rel_drop");
        if (rc) return rc;
        if (dkv2) return launch(attn32_bwd_dkv_rel2_kernel<true, 1>, smem_dkv(true, 1), p, st, "attn32_bwd_dkv_rel2_drop");
        return launch(attn32_bwd_dkv_kernel<true, true, 2>, smem_dkv(true, 2), p, st, "attn32_bwd_dkv_rel_drop");
    }
    rc = dq2 ? launch(attn32_bwd_dq_rel2_kernel<false>, smem_dq_rel2(), p, st, "attn32_bwd_dq_rel2")
             : launch(attn32_bwd_dq_kernel<true, false, 1>, smem_dq(true, 1), p, st, "attn32_bwd_dq_rel");
    if (rc) return rc;
    if (dkv2) return launch(attn32_bwd_dkv_rel2_kernel<false, 1>, smem_dkv(true, 1), p, st, "attn32_bwd_dkv_rel2");
    return launch(attn32_bwd_dkv_kernel<true, false, 2>, smem_dkv(true, 2), p, st, "attn32_bwd_dkv_rel");
}
