// block_wide.hip -- the block-tail chain kernels with 64 rows per workgroup.
//
// Why: a chain kernel is bound by the 64 B/clk a CU takes from L2 -- the block's whole weight set (1.25 MB) passes through that
// port once per WORKGROUP whatever its row count -- and in the training step the chip is shared by three chains, so what counts
// is CU-time per row.  Here every weight fragment a wave streams feeds TWO MFMAs (one per 32-row tile): half the CU-time per row
// of block_fused.hip / block_lean.hip at about the same latency per launch.
//
// Structure (ownership as in block_lean.hip): 4 waves, one per SIMD (512 registers); wave w owns output features 64 w .. 64 w + 63
// of every link and hidden tile 4 r + w of round r; the round's four activated hidden tiles are exchanged through a
// double-buffered LDS tile (one barrier per round); no cross-wave partial sums.  Different from block_lean.hip:
//  * the row tiles' activations (LN(x1) / dy / dx1) sit in registers as B fragments (2 x 16 x 4 registers);
//  * with one wave per SIMD nobody else hides a wave's VALU work, so the activation of round r is interleaved BY HAND
//    (`sched_barrier` fenced steps) with the MFMAs of round r + 1's first product AND of round r - 1's second product: the
//    second product lags one round, and the weight stream is packed in that order (blockpack.py, "wide");
//  * B fragments that come from LDS are requested one step ahead of the MFMAs that use them;
//  * every global access is a 16-byte-per-lane row-major access: inputs are staged through swizzled LDS tiles and read from there
//    in accumulator layout, outputs (x1, out, dx1, do) are written to LDS in accumulator layout and stored row-major (the
//    8-byte-per-lane accesses of accumulator layout cost a scattered 64-line request per instruction).
#include "block_common.h"

#define BW_ROWS 64
#define BW_HT_TILE 2560                         // one hidden tile: [32 rows][80 B]
// LDS carve, forward                           // backward
#define BW_OT 0                                 // o tiles 2 x [32][512], later y tiles 2 x [32][256]   | DT: dy tiles (32 KB), XT at 32768
#define BW_XT 65536                             // x0 -> x1 -> out tiles 2 x [32][256]                  | (do tiles alias DT + XT)
#define BW_HT (BW_XT + 32768)                   // 2 buffers x 4 hidden tiles x 2 row tiles
#define BW_STAT (BW_HT + 2 * 4 * 2 * BW_HT_TILE)
#define BW_BIAS (BW_STAT + 2 * 2 * 128 * 4)
#define BW_PAR (BW_BIAS + 4 * BF_MAX_F)
#define BW_TOTAL (BW_PAR + 4 * 4 * BF_D)        // 153 600 B

// per-row sums over the 256 features of NQ quantities each wave holds for its 64 features, both row tiles, ONE barrier
template <int NQ>
__device__ __forceinline__ void bw_rowsum(char* smem, int slot0, int wave, int lane, const float (&partial)[NQ][2], float (&total)[NQ][2]) {
    float* st = reinterpret_cast<float*>(smem + BW_STAT);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float v = partial[q][t];
            v += __shfl_xor(v, 32, 64);
            if (lane < 32) st[((slot0 + q) * 2 + t) * 128 + wave * 32 + lane] = v;
        }
    __syncthreads();
    const int m = lane & 31;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float* s = st + ((slot0 + q) * 2 + t) * 128 + m;
            total[q][t] = s[0] + s[32] + s[64] + s[96];
        }
}
// every wave publishes two per-row values per row tile (lanes of both halves hold the same value); all[q * 4 + w][t] = wave w's value q
__device__ __forceinline__ void bw_exchange(char* smem, int wave, int lane, const float (&mine)[2][2], float (&all)[8][2]) {
    float* st = reinterpret_cast<float*>(smem + BW_STAT);
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int t = 0; t < 2; ++t)
            if (lane < 32) st[(q * 2 + t) * 128 + wave * 32 + lane] = mine[q][t];
    __syncthreads();
    const int m = lane & 31;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int w = 0; w < 4; ++w) all[q * 4 + w][t] = st[(q * 2 + t) * 128 + w * 32 + m];
}
__device__ __forceinline__ bf16x8 bw_frag256(const char* tile, int m, int h, int ks) {
    return *reinterpret_cast<const bf16x8*>(tile + m * 512 + (((2 * ks + h) ^ (m & 15)) << 4));
}
__device__ __forceinline__ bf16x8 bw_frag512(const char* tile, int m, int h, int ks) {
    return *reinterpret_cast<const bf16x8*>(tile + m * 1024 + (((2 * ks + h) ^ (m & 15)) << 4));
}
__device__ __forceinline__ void bw_put_tile(char* tile, int m, int h, const bf16x8 (&hb)[2]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const bf16x8& s = hb[g >> 1];
        const int o = (g & 1) * 4;
        *reinterpret_cast<bf16x4*>(tile + m * 80 + (8 * g + 4 * h) * 2) = bf16x4{s[o], s[o + 1], s[o + 2], s[o + 3]};
    }
}
__device__ __forceinline__ bf16x8 bw_get_frag(const char* tile, int m, int h, int s) {
    return *reinterpret_cast<const bf16x8*>(tile + m * 80 + (16 * s + 8 * h) * 2);
}
// row-major global [rows][256] bf16 <-> the two swizzled [32][256] tiles at `tiles`: 8 chunks of 16 B per thread
__device__ __forceinline__ void bw_load_rows256(const bf16_t* src, int m0, int M, bf16x8 (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = i * 256 + threadIdx.x;
        const int r = q >> 5, ch = q & 31;
        v[i] = *reinterpret_cast<const bf16x8*>(src + (size_t)min(m0 + r, M - 1) * BF_D + 8 * ch);
    }
}
__device__ __forceinline__ void bw_rows256_to_lds(char* tiles, const bf16x8 (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = i * 256 + threadIdx.x;
        const int r = q >> 5, ch = q & 31;
        *reinterpret_cast<bf16x8*>(tiles + (r >> 5) * 16384 + (r & 31) * 512 + ((ch ^ (r & 15)) << 4)) = v[i];
    }
}
__device__ __forceinline__ void bw_store_rows256(const char* tiles, bf16_t* dst, int m0, int M) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = i * 256 + threadIdx.x;
        const int r = q >> 5, ch = q & 31;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(tiles + (r >> 5) * 16384 + (r & 31) * 512 + ((ch ^ (r & 15)) << 4));
        if (m0 + r < M) *reinterpret_cast<bf16x8*>(dst + (size_t)(m0 + r) * BF_D + 8 * ch) = v;
    }
}

// cold-weight touches (bf_touch_stream) with a compile-time trip count: a run-time loop of loads makes the compiler wait for ALL
// outstanding loads at the next use of any of them (the counter is unknown after the loop)
struct BwTouch { unsigned v[8]; };
__device__ __forceinline__ BwTouch bw_touch_stream(const void* stream, int total_frags) {
    const int part = (blockIdx.x >> 3) & 7;
    const int lines = total_frags;                     // one 128-byte line in eight, an eighth of them per workgroup
    const char* base = reinterpret_cast<const char*>(stream) + (size_t)part * lines * 128;
    BwTouch r;
#pragma unroll
    for (int i = 0; i < 8; ++i)                        // (total_frags <= 2048: F <= 2048, DI <= 512; surplus lanes re-touch the last line)
        r.v[i] = *reinterpret_cast<const unsigned*>(base + (size_t)min(i * 256 + (int)threadIdx.x, lines - 1) * 128);
    return r;                                          // (nobody reads the values before the kernel's end: no wait on these loads)
}
__device__ __forceinline__ unsigned bw_touch_fold(const BwTouch& r) {
    unsigned a = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) a |= r.v[i];
    return a;
}

// first product of a hidden tile: accn[t] += (16 streamed fragments, ring slots S0 .. S0+15) . (B fragments of the two [32][256]
// tiles at `tiles`, requested BW_AHEAD steps ahead of their MFMAs -- holding all of them in registers (128) does not fit beside
// the ring and the accumulators); every slot is re-requested for stream position + 32; side(ks) = activation work fenced behind
// this step's MFMAs
#define BW_AHEAD 3
template <int S0, class F>
__device__ __forceinline__ void bw_first(bf16x8 (&ring)[BF_RING], const bf16x8* nx, const char* tiles, int m, int h, f32x16 (&accn)[2], F&& side) {
    bf16x8 bq[BW_AHEAD + 1][2];
#pragma unroll
    for (int ks = 0; ks < BW_AHEAD; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t) bq[ks][t] = bw_frag256(tiles + t * 16384, m, h, ks);
#pragma unroll
    for (int ks = 0; ks < BF_KS; ++ks) {
        if (ks + BW_AHEAD < BF_KS) {
#pragma unroll
            for (int t = 0; t < 2; ++t) bq[(ks + BW_AHEAD) % (BW_AHEAD + 1)][t] = bw_frag256(tiles + t * 16384, m, h, ks + BW_AHEAD);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) accn[t] = mfma32(ring[S0 + ks], bq[ks % (BW_AHEAD + 1)][t], accn[t]);
        ring[S0 + ks] = nx[ks * 64];
        side(ks);
        if (ks & 1) __builtin_amdgcn_sched_barrier(0);
    }
}
// second product of a round: acc[t][c2] += (this wave's two feature tiles of the streamed weight, ring slots S0 .. S0+15 in order
// [k'][c2]) . (the round's four hidden tiles of row tile t from the exchange buffer `ht`: hidden tile q of row tile t at
// (2 q + t) BW_HT_TILE), B fragments requested one step ahead;  RELOAD: re-request the slots for stream positions + 32
template <int S0, bool RELOAD, class F>
__device__ __forceinline__ void bw_second(bf16x8 (&ring)[BF_RING], const bf16x8* nx, const char* ht, int m, int h, f32x16 (&acc)[2][2], F&& side) {
    bf16x8 hf[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) hf[0][t] = bw_get_frag(ht + t * BW_HT_TILE, m, h, 0);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (k + 1 < 8) {
#pragma unroll
            for (int t = 0; t < 2; ++t) hf[(k + 1) & 1][t] = bw_get_frag(ht + (((k + 1) >> 1) * 2 + t) * BW_HT_TILE, m, h, (k + 1) & 1);
        }
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            const int j = 2 * k + c2;
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t][c2] = mfma32(ring[S0 + j], hf[k & 1][t], acc[t][c2]);
            if (RELOAD) ring[S0 + j] = nx[j * 64];
        }
        side(k);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// the residual tiles and the parameters into LDS (requested at the start, needed only behind the output projection) + barrier
__device__ __forceinline__ void bw_stage_inputs(char* smem, int wave, int lane, const bf16x8 (&xv)[8], const f32x4& pv, const f32x4 (&b1v)[2], int F) {
    bw_rows256_to_lds(smem + BW_XT, xv);
    reinterpret_cast<f32x4*>(smem + BW_PAR)[wave * 64 + lane] = pv;
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (k * 256 + (int)threadIdx.x < F / 4) reinterpret_cast<f32x4*>(smem + BW_BIAS)[k * 256 + threadIdx.x] = b1v[k];
    __syncthreads();
}

struct WideFwd {
    int M;
    const bf16_t* o; int ldo;
    const bf16_t* x0;
    const bf16x8* Wst; int wave_frags;
    const float* bo;
    bf16_t* x1;
    const float* gamma; const float* beta; float eps;
    const float* b1; int F;
    const float* b2;
    bf16_t* z;
    float* mean; float* rstd;
    bf16_t* out;
};

// AR = DI / 256 (0 = no output projection, x1 is the input).  Stream per wave w (blockpack.py, "wide"): projection fragments of
// its feature tiles 2w, 2w+1 in order [ks][c2] (DI / 8 of them); then, with G1(r) = W1 of hidden tile 4 r + w (16 fragments [ks])
// and G2(r) = W2 of its two feature tiles over round r's 128 hidden units (16 fragments [k'][c2]):
//   G1(0), G1(1), { G1(r + 1), G2(r - 1) : r = 1 .. nr - 2 }, G2(nr - 2), G2(nr - 1)      (nr = F / 128 >= 2: F = 128 goes to block_lean.hip)
// Groups alternate between the ring halves (slots 0..15 / 16..31); the last two groups are not re-requested.
template <int ACT, int AR>
__global__ __launch_bounds__(256, 1) void block_tail_wide_fwd_kernel(WideFwd a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * BW_ROWS;
    int row[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) row[t] = min(m0 + 32 * t + m, a.M - 1);
    BF_STAMP(0);

    // ---- requests in the order their data is needed (a CU takes 64 B/clk from L2 and returns loads in order: whatever is requested
    // in front of the o tiles delays the first MFMA): o tiles, the ring, then x0 / the parameters (used after the projection), touches
    constexpr int NO = AR > 0 ? 8 * AR : 1;            // 16-byte chunks of o per thread (64 rows x DI / 8 chunks)
    bf16x8 ov[NO];
    if (AR > 0) {
        constexpr int CPR = AR > 0 ? 32 * AR : 1;      // chunks per row
#pragma unroll
        for (int i = 0; i < NO; ++i) {
            const int q = i * 256 + threadIdx.x;
            const int r = q / CPR, ch = q % CPR;
            ov[i] = *reinterpret_cast<const bf16x8*>(a.o + (size_t)min(m0 + r, a.M - 1) * a.ldo + 8 * ch);
        }
    }
    const bf16x8* nx = a.Wst + (size_t)wave * a.wave_frags * 64 + lane;
    bf16x8 ring[BF_RING];
#pragma unroll
    for (int i = 0; i < BF_RING; ++i) ring[i] = nx[i * 64];
    nx += BF_RING * 64;
    bf16x8 xv[8];
    bw_load_rows256(AR > 0 ? a.x0 : a.x1, m0, a.M, xv);
    f32x4 pv = {0.f, 0.f, 0.f, 0.f}, b1v[2];
    {
        const float* srcp = wave == 0 ? a.bo : wave == 1 ? a.gamma : wave == 2 ? a.beta : a.b2;
        if (srcp != nullptr) pv = reinterpret_cast<const f32x4*>(srcp)[lane];
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (k * 256 + (int)threadIdx.x < a.F / 4) b1v[k] = reinterpret_cast<const f32x4*>(a.b1)[k * 256 + threadIdx.x];
    }
    BwTouch touched;
    if (BF_TOUCH) touched = bw_touch_stream(a.Wst, 4 * a.wave_frags);
    if (AR > 0) {
        constexpr int CPR = AR > 0 ? 32 * AR : 1;
#pragma unroll
        for (int i = 0; i < NO; ++i) {
            const int q = i * 256 + threadIdx.x;
            const int r = q / CPR, ch = q % CPR;
            *reinterpret_cast<bf16x8*>(smem + BW_OT + (r >> 5) * 32768 + (r & 31) * 1024 + ((ch ^ (r & 15)) << 4)) = ov[i];
        }
        __syncthreads();                               // o tiles are in LDS
    }
    const float* pbo = reinterpret_cast<const float*>(smem + BW_PAR), *pgam = pbo + BF_D, *pbet = pbo + 2 * BF_D, *pb2 = pbo + 3 * BF_D;
    BF_STAMP(1);

    bf16x4 xb[2][2][4];                                // this wave's 64 features of x1, both row tiles (accumulator layout)
    if (AR > 0) {
        // ---- x1 = x0 + o Wo^T + bo for this wave's 64 features: no k split, no exchange
        f32x16 acc[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) { acc[t][0] = zero16(); acc[t][1] = zero16(); }
        bf16x8 of[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) of[0][t] = bw_frag512(smem + BW_OT + t * 32768, m, h, 0);
#pragma unroll
        for (int rr = 0; rr < AR; ++rr) {
#pragma unroll
            for (int kl = 0; kl < 16; ++kl) {
                const int ks = 16 * rr + kl;
                if (ks + 1 < 16 * AR) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) of[(ks + 1) & 1][t] = bw_frag512(smem + BW_OT + t * 32768, m, h, ks + 1);
                }
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    const int j = 2 * kl + c2;
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[t][c2] = mfma32(ring[j], of[ks & 1][t], acc[t][c2]);
                    ring[j] = nx[j * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            nx += BF_RING * 64;
        }
        BF_STAMP(2);
        bw_stage_inputs(smem, wave, lane, xv, pv, b1v, a.F);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = 64 * wave + 32 * c2 + 8 * g + 4 * h;
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(pbo + c);
                    char* px = smem + BW_XT + t * 16384 + bf_tile_off(m, c);
                    const bf16x4 x0v = *reinterpret_cast<const bf16x4*>(px);
#pragma unroll
                    for (int i = 0; i < 4; ++i) xb[t][c2][g][i] = (bf16_t)(acc[t][c2][4 * g + i] + bb[i] + (float)x0v[i]);
                    *reinterpret_cast<bf16x4*>(px) = xb[t][c2][g];
                }
    } else {
        bw_stage_inputs(smem, wave, lane, xv, pv, b1v, a.F);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    xb[t][c2][g] = *reinterpret_cast<const bf16x4*>(smem + BW_XT + t * 16384 + bf_tile_off(m, 64 * wave + 32 * c2 + 8 * g + 4 * h));
    }
    BF_STAMP(3);
    // ---- LayerNorm -> y tiles (the o tiles' place).  Statistics with ONE exchange: every wave's mean and centred sum of squares over
    // its own 64 features (two passes in registers), combined as in a parallel variance: M2 = sum M2_w + 64 sum (mean_w - mean)^2
    float st[2][2], tot[8][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float sw = 0.f;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) sw += (float)xb[t][c2][g][i];
        sw += __shfl_xor(sw, 32, 64);
        const float mw = sw * (1.f / 64.f);
        float qw = 0.f;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float d = (float)xb[t][c2][g][i] - mw; qw += d * d; }
        qw += __shfl_xor(qw, 32, 64);
        st[0][t] = mw; st[1][t] = qw;
    }
    bw_exchange(smem, wave, lane, st, tot);             // (barrier: every wave's x1 is in the x tiles, nobody reads the o tiles any more)
    if (AR > 0) bw_store_rows256(smem + BW_XT, a.x1, m0, a.M);
    float mean[1][2], var[1][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float mu = 0.25f * (tot[0][t] + tot[1][t] + tot[2][t] + tot[3][t]);
        float m2 = tot[4][t] + tot[5][t] + tot[6][t] + tot[7][t];
#pragma unroll
        for (int w = 0; w < 4; ++w) { const float d = tot[w][t] - mu; m2 = fmaf(64.f * d, d, m2); }
        mean[0][t] = mu;
        var[0][t] = m2;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float rstd = rsqrtf(var[0][t] * (1.f / BF_D) + a.eps);
        if (wave == 0 && lane < 32 && m0 + 32 * t + m < a.M) { a.mean[row[t]] = mean[0][t]; a.rstd[row[t]] = rstd; }
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 64 * wave + 32 * c2 + 8 * g + 4 * h;
                const f32x4 gg = *reinterpret_cast<const f32x4*>(pgam + c);
                const f32x4 be = *reinterpret_cast<const f32x4*>(pbet + c);
                bf16x4 y;
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] = (bf16_t)(((float)xb[t][c2][g][i] - mean[0][t]) * rstd * gg[i] + be[i]);
                *reinterpret_cast<bf16x4*>(smem + BW_OT + t * 16384 + bf_tile_off(m, c)) = y;
            }
    }
    __syncthreads();
    const char* const YT = smem + BW_OT;               // (the y tiles stay here: the first products read their B fragments from them)
    BF_STAMP(4);

    // ---- feed-forward in rounds of four hidden tiles (wave w: tile 4 r + w); this wave's 64 output features in acc2
    const int nr = a.F / 128;
    const float* b1s = reinterpret_cast<const float*>(smem + BW_BIAS);
    f32x16 acc2[2][2], acc1[2], accn[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { acc2[t][0] = zero16(); acc2[t][1] = zero16(); acc1[t] = bf_bias_init(b1s, wave, h); }
    auto none = [](int) __attribute__((always_inline)) {};
    // z of whole 64-row groups (the entry point checks that z is given and holds them)
    bf16x8* zp = reinterpret_cast<bf16x8*>(a.z) + ((size_t)(blockIdx.x * 2 * (a.F / 32) + wave) * 64 + lane) * 2;
    const size_t ztile = (size_t)(a.F / 32) * 128;      // bf16x8 units between the two row tiles of a group
    // activation of element e of round `rc` (e >> 4 = row tile, e & 15 = accumulator register), called in increasing e: the
    // pre-activation goes to z and the activated value into the round's exchange buffer as soon as a 16- / 8-byte piece is complete
    // (buffer rc & 1 is free: the barrier that ended round rc - 1 came after every wave's reads of round rc - 2)
    int rc = 0;
    bf16x8 zcur[2];
    bf16x4 hcur[2];
    auto act = [&](int e) __attribute__((always_inline)) {
        const int t = e >> 4, i = e & 15;
        float g, dg;
#ifdef BW_DBG_NOACT
        g = acc1[t][i]; dg = g;
#else
        bf_gelu2<ACT>(acc1[t][i], g, dg);
#endif
        zcur[t][i & 7] = (bf16_t)dg;
        hcur[t][i & 3] = (bf16_t)g;
        if ((i & 3) == 3)
            *reinterpret_cast<bf16x4*>(smem + BW_HT + ((rc & 1) * 8 + wave * 2 + t) * BW_HT_TILE + m * 80 + (8 * (i >> 2) + 4 * h) * 2) = hcur[t];
        if ((i & 7) == 7) zp[t * ztile + (size_t)rc * 4 * 128 + (i >> 3)] = zcur[t];
    };
    auto finish = [&](int r) __attribute__((always_inline)) { __syncthreads(); };
    bw_first<0>(ring, nx, YT, m, h, acc1, none);              // G1(0)
    nx += 16 * 64;
    BF_STAMP(5);
    {                                                   // (nr >= 2: checked by the entry point)
        // round 0: its activation beside G1(1)
#pragma unroll
        for (int t = 0; t < 2; ++t) accn[t] = bf_bias_init(b1s, 4 + wave, h);
        bw_first<16>(ring, nx, YT, m, h, accn, [&](int ks) __attribute__((always_inline)) { act(2 * ks); act(2 * ks + 1); });
        nx += 16 * 64;
        finish(0);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc1[t] = accn[t];
        for (int r = 1; r + 1 < nr; ++r) {
            rc = r;
            BF_STAMP(8 + 2 * (r - 1));
            // round r: row tile 0's activation beside G1(r + 1), row tile 1's beside G2(r - 1)
#pragma unroll
            for (int t = 0; t < 2; ++t) accn[t] = bf_bias_init(b1s, 4 * (r + 1) + wave, h);
            bw_first<0>(ring, nx, YT, m, h, accn, [&](int ks) __attribute__((always_inline)) { if (ks & 1) { act(ks - 1); act(ks); } });
            BF_STAMP(9 + 2 * (r - 1));
            bw_second<16, true>(ring, nx + 16 * 64, smem + BW_HT + ((r - 1) & 1) * 8 * BW_HT_TILE, m, h, acc2,
                                [&](int k) __attribute__((always_inline)) { act(16 + 2 * k); act(17 + 2 * k); });
            nx += BF_RING * 64;
            finish(r);
#pragma unroll
            for (int t = 0; t < 2; ++t) acc1[t] = accn[t];
        }
        BF_STAMP(6);
        rc = nr - 1;
        // last round: its activation beside G2(nr - 2)
        bw_second<0, false>(ring, nx, smem + BW_HT + ((nr - 2) & 1) * 8 * BW_HT_TILE, m, h, acc2,
                            [&](int k) __attribute__((always_inline)) { act(4 * k); act(4 * k + 1); act(4 * k + 2); act(4 * k + 3); });
        finish(nr - 1);
    }
    bw_second<16, false>(ring, nx, smem + BW_HT + ((nr - 1) & 1) * 8 * BW_HT_TILE, m, h, acc2, none);     // G2(nr - 1)
    BF_STAMP(7);
    // ---- out = x1 + h W2^T + b2: into the x tiles (each lane over its own x1 values), then row-major to global
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 64 * wave + 32 * c2 + 8 * g + 4 * h;
                const f32x4 bb = *reinterpret_cast<const f32x4*>(pb2 + c);
                char* px = smem + BW_XT + t * 16384 + bf_tile_off(m, c);
                const bf16x4 x1r = *reinterpret_cast<const bf16x4*>(px);
                bf16x4 yo;
#pragma unroll
                for (int i = 0; i < 4; ++i) yo[i] = (bf16_t)(acc2[t][c2][4 * g + i] + bb[i] + (float)x1r[i]);
                *reinterpret_cast<bf16x4*>(px) = yo;
            }
    __syncthreads();
    bw_store_rows256(smem + BW_XT, a.out, m0, a.M);
    BF_STAMP(10 + 16);
    if (BF_TOUCH && bw_touch_fold(touched) == 0x7fc07fc1u && a.M < 0) a.mean[0] = 0.f;      // (keeps the prefetch loads alive; never true)
}

struct WideBwd {
    int M;
    const bf16_t* x1; const bf16_t* dy;
    const float* gamma; const float* mean; const float* rstd;
    const bf16_t* z;
    const bf16x8* Wst; int wave_frags; int F;
    bf16_t* dx1;
    bf16_t* dout; int lddo;
    const bf16_t* ao; const bf16_t* ao_lo; int ldao; float* delta; int T;      // (optional: delta for the attention backward, cvft.h)
};

// Stream per wave w ("wide"): with H1(r) = W2^T of hidden tile 4 r + w (16 fragments [ks]) and H2(r) = W1^T of its two feature
// tiles over round r's 128 hidden units (16 fragments [k'][c2]):
//   H1(0), H1(1), { H1(r + 1), H2(r - 1) : r = 1 .. nr - 2 }, H2(nr - 2), H2(nr - 1);  then (CR = DI / 256 > 0) Wo^T of its DI / 4
// output features in order [ks][f] (the last two H groups are re-requested only when this link follows).
template <int ACT, int CR>
__global__ __launch_bounds__(256, 1) void block_tail_wide_bwd_kernel(WideBwd a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * BW_ROWS;
    char* const DT = smem;                              // dy tiles
    char* const XT = smem + 32768;                      // x1 tiles, then dx1 tiles
    const int nr = a.F / 128;

    BF_STAMP(16);
    // dy and x1 tiles, gamma, the row statistics, the first z tiles, the ring, the cold-weight touches
    bf16x8 dv[8], xv[8];
    bw_load_rows256(a.dy, m0, a.M, dv);
    bw_load_rows256(a.x1, m0, a.M, xv);
    if (wave == 0) reinterpret_cast<f32x4*>(smem + BW_PAR)[lane] = reinterpret_cast<const f32x4*>(a.gamma)[lane];
    float mean[2], rstd[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int r = min(m0 + 32 * t + m, a.M - 1);
        mean[t] = a.mean[r];
        rstd[t] = a.rstd[r];
    }
    const bf16x8* zp = reinterpret_cast<const bf16x8*>(a.z) + ((size_t)(blockIdx.x * 2 * (a.F / 32) + wave) * 64 + lane) * 2;
    const size_t ztile = (size_t)(a.F / 32) * 128;
    bf16x8 zs[2][2], zn[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { zs[t][0] = zp[t * ztile]; zs[t][1] = zp[t * ztile + 1]; }
    const bf16x8* nx = a.Wst + (size_t)wave * a.wave_frags * 64 + lane;
    bf16x8 ring[BF_RING];
#pragma unroll
    for (int i = 0; i < BF_RING; ++i) ring[i] = nx[i * 64];
    nx += BF_RING * 64;
    BwTouch touched;
    if (BF_TOUCH) touched = bw_touch_stream(a.Wst, 4 * a.wave_frags);
    bw_rows256_to_lds(DT, dv);
    bw_rows256_to_lds(XT, xv);
    const float* pgam = reinterpret_cast<const float*>(smem + BW_PAR);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) { BF_PIN(mean[t]); BF_PIN(rstd[t]); }
    BF_STAMP(17);
    f32x16 accd[2][2], accg[2], accn[2];
    int rc = 0;
    bf16x4 hcur[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { accd[t][0] = zero16(); accd[t][1] = zero16(); accg[t] = zero16(); }
    auto none = [](int) __attribute__((always_inline)) {};
    auto act = [&](int e) __attribute__((always_inline)) {
        const int t = e >> 4, i = e & 15;
        hcur[t][i & 3] = (bf16_t)(accg[t][i] * (float)zs[t][i >> 3][i & 7]);       // (z holds gelu'(z): see bf_gelu2)
        if ((i & 3) == 3)
            *reinterpret_cast<bf16x4*>(smem + BW_HT + ((rc & 1) * 8 + wave * 2 + t) * BW_HT_TILE + m * 80 + (8 * (i >> 2) + 4 * h) * 2) = hcur[t];
    };
    auto zload = [&](int r) __attribute__((always_inline)) {     // next round's pre-activations (consumed a round later)
#pragma unroll
        for (int t = 0; t < 2; ++t) { zn[t][0] = zp[t * ztile + (size_t)r * 4 * 128]; zn[t][1] = zp[t * ztile + (size_t)r * 4 * 128 + 1]; }
    };
    auto finish = [&](int r, bool more) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
            if (more) { zs[t][0] = zn[t][0]; zs[t][1] = zn[t][1]; accg[t] = accn[t]; }
        __syncthreads();
    };
    bw_first<0>(ring, nx, DT, m, h, accg, none);              // H1(0)
    nx += 16 * 64;
    BF_STAMP(18);
    {                                                   // (nr >= 2: checked by the entry point)
        zload(1);
#pragma unroll
        for (int t = 0; t < 2; ++t) accn[t] = zero16();
        bw_first<16>(ring, nx, DT, m, h, accn, [&](int ks) __attribute__((always_inline)) { act(2 * ks); act(2 * ks + 1); });
        nx += 16 * 64;
        finish(0, true);
        BF_STAMP(19);
        for (int r = 1; r + 1 < nr; ++r) {
            rc = r;
            zload(r + 1);
#pragma unroll
            for (int t = 0; t < 2; ++t) accn[t] = zero16();
            bw_first<0>(ring, nx, DT, m, h, accn, [&](int ks) __attribute__((always_inline)) { if (ks & 1) { act(ks - 1); act(ks); } });
            bw_second<16, true>(ring, nx + 16 * 64, smem + BW_HT + ((r - 1) & 1) * 8 * BW_HT_TILE, m, h, accd,
                                [&](int k) __attribute__((always_inline)) { act(16 + 2 * k); act(17 + 2 * k); });
            nx += BF_RING * 64;
            finish(r, true);
        }
        BF_STAMP(20);
        rc = nr - 1;
        bw_second<0, (CR > 0)>(ring, nx, smem + BW_HT + ((nr - 2) & 1) * 8 * BW_HT_TILE, m, h, accd,
                               [&](int k) __attribute__((always_inline)) { act(4 * k); act(4 * k + 1); act(4 * k + 2); act(4 * k + 3); });
        nx += 16 * 64;
        finish(nr - 1, false);
    }
    bw_second<16, (CR > 0)>(ring, nx, smem + BW_HT + ((nr - 1) & 1) * 8 * BW_HT_TILE, m, h, accd, none);      // H2(nr - 1)
    nx += 16 * 64;

    BF_STAMP(21);
    // ---- LayerNorm backward + residual branch for this wave's 64 features; x1 and dy from the tiles in accumulator layout
    float sp[2][2], sm[2][2];
    bf16x4 xr[2][2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        sp[0][t] = 0.f; sp[1][t] = 0.f;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 64 * wave + 32 * c2 + 8 * g + 4 * h;
                xr[t][c2][g] = *reinterpret_cast<const bf16x4*>(XT + t * 16384 + bf_tile_off(m, c));
                const f32x4 gg = *reinterpret_cast<const f32x4*>(pgam + c);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = 4 * g + i;
                    accd[t][c2][e] *= gg[i];            // gamma . d(normalised)
                    sp[0][t] += accd[t][c2][e];
                    sp[1][t] += accd[t][c2][e] * (((float)xr[t][c2][g][i] - mean[t]) * rstd[t]);
                }
            }
    }
    BF_STAMP(22);
    bw_rowsum<2>(smem, 0, wave, lane, sp, sm);
    BF_STAMP(23);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float m1 = sm[0][t] * (1.f / BF_D), m2 = sm[1][t] * (1.f / BF_D);
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 64 * wave + 32 * c2 + 8 * g + 4 * h;
                const bf16x4 dr = *reinterpret_cast<const bf16x4*>(DT + t * 16384 + bf_tile_off(m, c));
                bf16x4 dx;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = 4 * g + i;
                    dx[i] = (bf16_t)((float)dr[i] + rstd[t] * (accd[t][c2][e] - m1 - (((float)xr[t][c2][g][i] - mean[t]) * rstd[t]) * m2));
                }
                *reinterpret_cast<bf16x4*>(XT + t * 16384 + bf_tile_off(m, c)) = dx;      // (over this lane's own x1 values)
            }
    }
    __syncthreads();
    BF_STAMP(24);
    bw_store_rows256(XT, a.dx1, m0, a.M);
    if (BF_TOUCH && bw_touch_fold(touched) == 0x7fc07fc1u && a.M < 0) a.dx1[0] = (bf16_t)0.f;  // (keeps the prefetch loads alive; never true)
    if (CR == 0) return;
    // ---- do = dx1 Wo for this wave's DI / 4 output features (2 CR tiles of 32), stream order [ks][f]; ring positions continue at slot 0
    {
        constexpr int NF = CR > 0 ? 2 * CR : 1;        // feature tiles per wave
        f32x16 acc[2][NF];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int f = 0; f < NF; ++f) acc[t][f] = zero16();
        constexpr int CPR = CR > 0 ? 32 * CR : 32;     // 16-byte chunks per row
        constexpr int NIT = 8 * (CR > 0 ? CR : 1);
        // delta for the attention backward that consumes dout (optional, cvft.h): this thread's chunks of the attention output (a chunk
        // = 8 columns; 8 consecutive lanes = one head's 64), requested in front of the projection's products: they arrive under the
        // first half of them (requested any earlier -- in front of the LayerNorm backward -- the kernel spills 40 registers)
        const bool wdel = a.delta != nullptr;
        bf16x8 ob[NIT], lb[NIT];
        if (wdel) {
            const bf16_t* aol = a.ao_lo ? a.ao_lo : a.ao;
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int q = i * 256 + threadIdx.x;
                const int r = min(m0 + q / CPR, a.M - 1), ch = q % CPR;
                ob[i] = *reinterpret_cast<const bf16x8*>(a.ao + (size_t)r * a.ldao + 8 * ch);
                lb[i] = *reinterpret_cast<const bf16x8*>(aol + (size_t)r * a.ldao + 8 * ch);
            }
        }
        const float lo_w = a.ao_lo ? 1.f : 0.f;
        bf16x8 xq[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) xq[0][t] = bw_frag256(XT + t * 16384, m, h, 0);
#pragma unroll
        for (int ks = 0; ks < BF_KS; ++ks) {
            if (ks + 1 < BF_KS) {
#pragma unroll
                for (int t = 0; t < 2; ++t) xq[(ks + 1) & 1][t] = bw_frag256(XT + t * 16384, m, h, ks + 1);
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int p = ks * NF + f;
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t][f] = mfma32(ring[p % BF_RING], xq[ks & 1][t], acc[t][f]);
                if (p + BF_RING < 16 * NF) ring[p % BF_RING] = nx[(size_t)p * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        BF_STAMP(25);
        __syncthreads();                               // every wave has read its dx1 fragments and stored dx1: the tiles are free
        char* OUT = smem;                              // 2 x [32][DI] tiles, row pitch 2 DI bytes, 16-byte chunks swizzled by (row & 15)
        constexpr int PITCH = CR > 0 ? 512 * CR : 512;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bf16x4 d = {(bf16_t)acc[t][f][4 * g], (bf16_t)acc[t][f][4 * g + 1], (bf16_t)acc[t][f][4 * g + 2], (bf16_t)acc[t][f][4 * g + 3]};
                    const int c = 32 * (wave * NF + f) + 8 * g + 4 * h;
                    *reinterpret_cast<bf16x4*>(OUT + t * 32 * PITCH + m * PITCH + (((c >> 3) ^ (m & 15)) << 4) + ((c & 7) << 1)) = d;
                }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int q = i * 256 + threadIdx.x;
            const int r = q / CPR, ch = q % CPR;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(OUT + (r >> 5) * 32 * PITCH + (r & 31) * PITCH + ((ch ^ (r & 15)) << 4));
            if (m0 + r < a.M) *reinterpret_cast<bf16x8*>(a.dout + (size_t)(m0 + r) * a.lddo + 8 * ch) = v;
            if (wdel) {
                float part = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) part += (float)v[e] * ((float)ob[i][e] + lo_w * (float)lb[i][e]);
                part += __shfl_xor(part, 1);
                part += __shfl_xor(part, 2);
                part += __shfl_xor(part, 4);
                const int row = m0 + r;
                if ((ch & 7) == 0 && row < a.M) a.delta[((size_t)(row / a.T) * (CPR / 8) + (ch >> 3)) * a.T + row % a.T] = part;
            }
        }
        BF_STAMP(27);
    }
}

template <typename K>
static int bw_prepare(K kernel) {
    if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BW_TOTAL) != hipSuccess) {
        cvft_set_error("block_wide: cannot reserve %d bytes of LDS", BW_TOTAL);
        return -2;
    }
    return 0;
}
template <int ACT, int AR>
static int launch_wide_fwd(const WideFwd& a, hipStream_t st) {
    static int ready = 0;
    if (!ready) { if (bw_prepare(block_tail_wide_fwd_kernel<ACT, AR>)) return -2; ready = 1; }
    hipLaunchKernelGGL((block_tail_wide_fwd_kernel<ACT, AR>), dim3((a.M + BW_ROWS - 1) / BW_ROWS), dim3(256), BW_TOTAL, st, a);
    return 0;
}
template <int ACT, int CR>
static int launch_wide_bwd(const WideBwd& a, hipStream_t st) {
    static int ready = 0;
    if (!ready) { if (bw_prepare(block_tail_wide_bwd_kernel<ACT, CR>)) return -2; ready = 1; }
    hipLaunchKernelGGL((block_tail_wide_bwd_kernel<ACT, CR>), dim3((a.M + BW_ROWS - 1) / BW_ROWS), dim3(256), BW_TOTAL, st, a);
    return 0;
}

// called by cvft_block_tail_fwd / _bwd (block_fused.hip) when args.lean == 2; arguments are already checked there
int block_tail_wide_fwd_launch(const cvft_block_tail_args* p, int DI, void* stream) {
    WideFwd a;
    a.M = p->M; a.o = (const bf16_t*)p->o; a.ldo = p->ldo; a.x0 = (const bf16_t*)p->x0;
    a.Wst = (const bf16x8*)p->W_fwd; a.wave_frags = DI / 8 + p->F / 4; a.bo = p->bo; a.x1 = (bf16_t*)p->x1;
    a.gamma = p->gamma; a.beta = p->beta; a.eps = p->eps; a.b1 = p->b1; a.F = p->F; a.b2 = p->b2;
    a.z = (bf16_t*)p->z; a.mean = p->mean; a.rstd = p->rstd; a.out = (bf16_t*)p->out;
    const bool erf = p->act == CVFT_ACT_GELU_ERF;
    hipStream_t st = (hipStream_t)stream;
    if (DI == 0) return erf ? launch_wide_fwd<CVFT_ACT_GELU_ERF, 0>(a, st) : launch_wide_fwd<CVFT_ACT_GELU_TANH, 0>(a, st);
    if (DI == 256) return erf ? launch_wide_fwd<CVFT_ACT_GELU_ERF, 1>(a, st) : launch_wide_fwd<CVFT_ACT_GELU_TANH, 1>(a, st);
    return erf ? launch_wide_fwd<CVFT_ACT_GELU_ERF, 2>(a, st) : launch_wide_fwd<CVFT_ACT_GELU_TANH, 2>(a, st);
}
int block_tail_wide_bwd_launch(const cvft_block_tail_bwd_args* p, int DI, void* stream) {
    WideBwd a;
    a.M = p->M; a.x1 = (const bf16_t*)p->x1; a.dy = (const bf16_t*)p->dy; a.gamma = p->gamma; a.mean = p->mean; a.rstd = p->rstd;
    a.z = (const bf16_t*)p->z; a.Wst = (const bf16x8*)p->W_bwd; a.wave_frags = p->F / 4 + p->DI / 8; a.F = p->F;
    a.dx1 = (bf16_t*)p->dx1; a.dout = (bf16_t*)p->dout; a.lddo = p->lddo;
    // (the optional delta output: checked by the caller, cvft_block_tail_bwd)
    a.ao = (const bf16_t*)p->attn_o; a.ao_lo = (const bf16_t*)p->attn_o_lo; a.ldao = p->ldao; a.delta = p->delta; a.T = p->T > 0 ? p->T : 1;
    const bool erf = p->act == CVFT_ACT_GELU_ERF;
    hipStream_t st = (hipStream_t)stream;
    if (DI == 0) return erf ? launch_wide_bwd<CVFT_ACT_GELU_ERF, 0>(a, st) : launch_wide_bwd<CVFT_ACT_GELU_TANH, 0>(a, st);
    if (DI == 256) return erf ? launch_wide_bwd<CVFT_ACT_GELU_ERF, 1>(a, st) : launch_wide_bwd<CVFT_ACT_GELU_TANH, 1>(a, st);
    return erf ? launch_wide_bwd<CVFT_ACT_GELU_ERF, 2>(a, st) : launch_wide_bwd<CVFT_ACT_GELU_TANH, 2>(a, st);
}
