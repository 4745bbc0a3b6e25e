// block_qkv_wide.hip -- the first half of the estimator block (block_qkv.hip: norm1 + LoRA q|k|v projection, and its backward) with
// 64 rows per workgroup: every streamed weight fragment feeds two MFMAs (one per 32-row tile), half the CU-time per row at the
// same bytes per workgroup (the tail's counterpart is block_wide.hip).  Same math, masks and operands as
// block_qkv.hip; what differs:
//   forward  -- x comes in and Y goes out as 16-byte-per-lane row-major accesses through LDS tiles (Y: 196 KB per workgroup, staged
//               per wave in pairs of n-tiles = whole 128-byte lines); LN(x) of both row tiles lives in registers as B fragments.
//   backward -- wave w owns 64 output features of dy for both row tiles and walks ALL of 3N (block_qkv.hip splits 3N over the
//               waves and sums 128 accumulator registers per wave through LDS: twice that does not fit); dY is staged through a
//               double-buffered LDS tile in chunks of 256 columns (row-major loads one chunk ahead), so every dY fragment is read
//               from LDS by the four waves and fetched from memory once.  The weight stream is re-packed for that order
//               (blockpack.py, W_bwd_wide: wave w = its two feature tiles, [ks][c2]).
#include "block_common.h"

#define QW_ROWS 64
// LDS carve
#define QW_XT 0                                 // x tiles 2 x [32][256] swizzled (backward: x, then dx)
#define QW_YT 32768                             // forward: LN(x) tiles;  backward: dres tiles
#define QW_STG 65536                            // 64 KB: forward: partial sums of the rank-side product, then per-wave Y staging;
                                                //        backward: the two dY chunk buffers, then the V partial sums
#define QW_STAT (QW_STG + 65536)                // 2 quantities x 2 row tiles x 4 waves x 32 rows
#define QW_BIAS (QW_STAT + 2048)                // 3N floats
#define QW_PAR (QW_BIAS + 6144)                 // gamma | beta
#define QW_TOTAL (QW_PAR + 2048)                // 141 312 B
#define QW_YPITCH 144                           // Y staging: [64 rows][128 B + 16]

__device__ __forceinline__ bf16x8 qw_mask8(bf16x8 v, unsigned long long key, unsigned long long e0, unsigned thr) {
    bool k0[4], k1[4];
    cvft_keep4(key, e0 >> 2, thr, k0);
    cvft_keep4(key, (e0 >> 2) + 1, thr, k1);
    uint4 u = *reinterpret_cast<uint4*>(&v);
    u.x &= (k0[0] ? 0x0000ffffu : 0u) | (k0[1] ? 0xffff0000u : 0u);
    u.y &= (k0[2] ? 0x0000ffffu : 0u) | (k0[3] ? 0xffff0000u : 0u);
    u.z &= (k1[0] ? 0x0000ffffu : 0u) | (k1[1] ? 0xffff0000u : 0u);
    u.w &= (k1[2] ? 0x0000ffffu : 0u) | (k1[3] ? 0xffff0000u : 0u);
    return *reinterpret_cast<bf16x8*>(&u);
}
__device__ __forceinline__ bf16x8 qw_frag256(const char* tile, int m, int h, int ks) {
    return *reinterpret_cast<const bf16x8*>(tile + m * 512 + (((2 * ks + h) ^ (m & 15)) << 4));
}
// row-major global [rows][ld] bf16, 256 columns from column c0 <-> the two swizzled [32][256] tiles: 8 chunks of 16 B per thread
__device__ __forceinline__ void qw_load_rows256(const bf16_t* src, size_t ld, int c0, int m0, int M, bf16x8 (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = i * 256 + threadIdx.x;
        const int r = q >> 5, ch = q & 31;
        v[i] = *reinterpret_cast<const bf16x8*>(src + (size_t)min(m0 + r, M - 1) * ld + c0 + 8 * ch);
    }
}
__device__ __forceinline__ void qw_rows256_to_lds(char* tiles, const bf16x8 (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = i * 256 + threadIdx.x;
        const int r = q >> 5, ch = q & 31;
        *reinterpret_cast<bf16x8*>(tiles + (r >> 5) * 16384 + (r & 31) * 512 + ((ch ^ (r & 15)) << 4)) = v[i];
    }
}
__device__ __forceinline__ void qw_store_rows256(const char* tiles, bf16_t* dst, int m0, int M) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = i * 256 + threadIdx.x;
        const int r = q >> 5, ch = q & 31;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(tiles + (r >> 5) * 16384 + (r & 31) * 512 + ((ch ^ (r & 15)) << 4));
        if (m0 + r < M) *reinterpret_cast<bf16x8*>(dst + (size_t)(m0 + r) * BF_D + 8 * ch) = v;
    }
}
// every wave publishes two per-row values per row tile; all[q * 4 + w][t] = wave w's value q
__device__ __forceinline__ void qw_exchange(char* smem, int wave, int lane, const float (&mine)[2][2], float (&all)[8][2]) {
    float* st = reinterpret_cast<float*>(smem + QW_STAT);
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
struct QwTouch { unsigned v[4]; };
__device__ __forceinline__ QwTouch qw_touch_stream(const void* stream, int total_frags) {
    const int part = (blockIdx.x >> 3) & 7;
    const int lines = total_frags;                     // one 128-byte line in eight, an eighth of them per workgroup (768 fragments)
    const char* base = reinterpret_cast<const char*>(stream) + (size_t)part * lines * 128;
    QwTouch r;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        r.v[i] = *reinterpret_cast<const unsigned*>(base + (size_t)min(i * 256 + (int)threadIdx.x, lines - 1) * 128);
    return r;                                          // (folded only at the kernel's end: no wait on these loads)
}
__device__ __forceinline__ unsigned qw_touch_fold(const QwTouch& r) { return r.v[0] | r.v[1] | r.v[2] | r.v[3]; }

struct QwFwd {
    int M;
    const bf16_t* x;
    const float* gamma; const float* beta; float eps;
    float* mean; float* rstd;
    const bf16x8* Wst;
    const float* bias;
    const bf16_t* A; int lda;
    const bf16_t* Bb; int ldb;
    float alpha; float p; const long long* seed; unsigned sites[3];
    bf16_t* U; int ldu;
    bf16_t* xd[3];
    bf16_t* y_out;
    bf16_t* Y; int ldy;
};

template <bool DROP>
__global__ __launch_bounds__(256, 1) void block_qkv_wide_fwd_kernel(QwFwd a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * QW_ROWS;
    int row[2];
    bool rvalid[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { row[t] = min(m0 + 32 * t + m, a.M - 1); rvalid[t] = m0 + 32 * t + m < a.M; }

    // ---- requests in the order their data is needed: x tiles, small parameters, adapter operands, the ring, touches
    bf16x8 xv[8];
    qw_load_rows256(a.x, BF_D, 0, m0, a.M, xv);
    f32x4 pv = {0.f, 0.f, 0.f, 0.f}, bv[2];
    if (wave < 2) pv = reinterpret_cast<const f32x4*>(wave == 0 ? a.gamma : a.beta)[lane];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int idx = k * 256 + threadIdx.x;
        bv[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (idx < 384 && a.bias != nullptr) bv[k] = reinterpret_cast<const f32x4*>(a.bias)[idx];
    }
    // adapter A operand of this wave's four k-steps (rows 0..31 = q|k adapters, rows 32..47 = v adapter, clamped beyond)
    bf16x8 af01[4], afv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ks = 4 * wave + k;
        af01[k] = *reinterpret_cast<const bf16x8*>(a.A + (size_t)m * a.lda + 16 * ks + 8 * h);
        afv[k] = *reinterpret_cast<const bf16x8*>(a.A + (size_t)(32 + (m & 15)) * a.lda + 16 * ks + 8 * h);
    }
    // B_blk fragment of n-tile nt (chained k order: element j = rank 8 (j>>2) + 4 h + (j&3) of the tile's adapter); requested one pair
    // of tiles ahead of its use
    const int nt0 = 12 * wave;
    auto bfrag = [&](int nt) __attribute__((always_inline)) {
        const int t = nt / 16;
        const bf16_t* bp = a.Bb + (size_t)(32 * nt + m) * a.ldb + 16 * t + 4 * h;
        const bf16x4 lo = *reinterpret_cast<const bf16x4*>(bp), hi = *reinterpret_cast<const bf16x4*>(bp + 8);
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    bf16x8 bext[2] = {bfrag(nt0), bfrag(nt0 + 1)}, bnext[2];
    const bf16x8* nx = a.Wst + (size_t)nt0 * BF_KS * 64 + lane;     // (the packed stream is [48 n-tiles][16 k-steps])
    bf16x8 ring[BF_RING];
#pragma unroll
    for (int i = 0; i < BF_RING; ++i) ring[i] = nx[i * 64];
    nx += BF_RING * 64;
    QwTouch touched;
    if (BF_TOUCH) touched = qw_touch_stream(a.Wst, 768);
    qw_rows256_to_lds(smem + QW_XT, xv);
    if (wave < 2) reinterpret_cast<f32x4*>(smem + QW_PAR)[wave * 64 + lane] = pv;
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (k * 256 + (int)threadIdx.x < 384) reinterpret_cast<f32x4*>(smem + QW_BIAS)[k * 256 + threadIdx.x] = bv[k];
    const float* pgam = reinterpret_cast<const float*>(smem + QW_PAR), *pbet = pgam + BF_D;
    __syncthreads();

    // ---- LayerNorm (statistics with one exchange, as block_wide.hip) -> y tiles -> B fragments of both row tiles in registers
    {
        bf16x4 xb[2][2][4];
        float st[2][2], tot[8][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float sw = 0.f;
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    xb[t][c2][g] = *reinterpret_cast<const bf16x4*>(smem + QW_XT + t * 16384 + bf_tile_off(m, 64 * wave + 32 * c2 + 8 * g + 4 * h));
#pragma unroll
                    for (int i = 0; i < 4; ++i) sw += (float)xb[t][c2][g][i];
                }
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
        qw_exchange(smem, wave, lane, st, tot);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float mu = 0.25f * (tot[0][t] + tot[1][t] + tot[2][t] + tot[3][t]);
            float m2 = tot[4][t] + tot[5][t] + tot[6][t] + tot[7][t];
#pragma unroll
            for (int w = 0; w < 4; ++w) { const float d = tot[w][t] - mu; m2 = fmaf(64.f * d, d, m2); }
            const float rstd = rsqrtf(m2 * (1.f / BF_D) + a.eps);
            if (wave == 0 && lane < 32 && rvalid[t]) { a.mean[row[t]] = mu; a.rstd[row[t]] = rstd; }
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = 64 * wave + 32 * c2 + 8 * g + 4 * h;
                    const f32x4 gg = *reinterpret_cast<const f32x4*>(pgam + c);
                    const f32x4 be = *reinterpret_cast<const f32x4*>(pbet + c);
                    bf16x4 y;
#pragma unroll
                    for (int i = 0; i < 4; ++i) y[i] = (bf16_t)(((float)xb[t][c2][g][i] - mu) * rstd * gg[i] + be[i]);
                    *reinterpret_cast<bf16x4*>(smem + QW_YT + t * 16384 + bf_tile_off(m, c)) = y;
                }
        }
    }
    __syncthreads();
    bf16x8 yf[2][BF_KS];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < BF_KS; ++ks) yf[t][ks] = qw_frag256(smem + QW_YT + t * 16384, m, h, ks);

    // ---- rank-side product: U^T[r, row] = sum_k A_t[r, k] drop_t(y)[row, k]; wave w takes k-steps 4w .. 4w+3 of both row tiles,
    // the partials meet in LDS.  A B-fragment lane holds 8 consecutive elements of its row: the mask groups of element index e0.
    f32x16 u01[2] = {zero16(), zero16()}, uv[2] = {zero16(), zero16()};
    const float inv_keep = DROP ? 1.f / (1.f - a.p) : 1.f;
    {
        unsigned long long keys[3] = {0, 0, 0};
        unsigned thr = 0;
        if (DROP) {
#pragma unroll
            for (int t = 0; t < 3; ++t) keys[t] = cvft_drop_key(a.seed, a.sites[t]);
            thr = cvft_drop_thr(a.p);
        }
        const bf16x8 zf = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ks = 4 * wave + k;               // (runtime index: re-read this k-step's fragments from the LDS tiles)
            const bf16x8 aq = m < 16 ? af01[k] : zf, ak = m < 16 ? zf : af01[k], av = m < 16 ? afv[k] : zf;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const bf16x8 yk = qw_frag256(smem + QW_YT + rt * 16384, m, h, ks);
                const unsigned long long e0 = (unsigned long long)row[rt] * BF_D + 16 * ks + 8 * h;
                bf16x8 vm[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    vm[t] = DROP ? qw_mask8(yk, keys[t], e0, thr) : yk;
                    if (DROP && a.xd[t] != nullptr && rvalid[rt]) {
                        bf16x8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)vm[t][e] * inv_keep);
                        *reinterpret_cast<bf16x8*>(a.xd[t] + e0) = o;
                    }
                }
                if (!DROP && a.y_out != nullptr && rvalid[rt]) *reinterpret_cast<bf16x8*>(a.y_out + e0) = yk;
                u01[rt] = mfma32(aq, vm[0], u01[rt]);
                u01[rt] = mfma32(ak, vm[1], u01[rt]);
                uv[rt] = mfma32(av, vm[2], uv[rt]);
            }
        }
        f32x4* part = reinterpret_cast<f32x4*>(smem + QW_STG);         // [wave][4 tiles][4 g][64 lanes]
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                part[((wave * 4 + rt) * 4 + g) * 64 + lane] = f32x4{u01[rt][4 * g], u01[rt][4 * g + 1], u01[rt][4 * g + 2], u01[rt][4 * g + 3]};
                part[((wave * 4 + 2 + rt) * 4 + g) * 64 + lane] = f32x4{uv[rt][4 * g], uv[rt][4 * g + 1], uv[rt][4 * g + 2], uv[rt][4 * g + 3]};
            }
        __syncthreads();
        const float sc = a.alpha * inv_keep;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 s0 = part[((0 * 4 + rt) * 4 + g) * 64 + lane], s1 = part[((0 * 4 + 2 + rt) * 4 + g) * 64 + lane];
#pragma unroll
                for (int w = 1; w < 4; ++w) { s0 += part[((w * 4 + rt) * 4 + g) * 64 + lane]; s1 += part[((w * 4 + 2 + rt) * 4 + g) * 64 + lane]; }
#pragma unroll
                for (int i = 0; i < 4; ++i) { u01[rt][4 * g + i] = s0[i] * sc; uv[rt][4 * g + i] = s1[i] * sc; }
            }
        __syncthreads();                               // (the partial sums are read: the area becomes the Y staging)
    }
    // U (bf16, as stored) is what the rank extension multiplies -- and what backward's dB = dY^T U reads
    bf16x8 hb[2][3];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int i = 0; i < 8; ++i) { hb[rt][0][i] = (bf16_t)u01[rt][i]; hb[rt][1][i] = (bf16_t)u01[rt][8 + i]; hb[rt][2][i] = (bf16_t)uv[rt][i]; }
    if (wave < 2 && rvalid[wave]) {                    // (wave 0 stores row tile 0's U, wave 1 row tile 1's)
        const int rt = wave;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const bf16x4 o = {(bf16_t)u01[rt][4 * g], (bf16_t)u01[rt][4 * g + 1], (bf16_t)u01[rt][4 * g + 2], (bf16_t)u01[rt][4 * g + 3]};
            *reinterpret_cast<bf16x4*>(a.U + (size_t)row[rt] * a.ldu + 8 * g + 4 * h) = o;
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const bf16x4 o = {(bf16_t)uv[rt][4 * g], (bf16_t)uv[rt][4 * g + 1], (bf16_t)uv[rt][4 * g + 2], (bf16_t)uv[rt][4 * g + 3]};
            *reinterpret_cast<bf16x4*>(a.U + (size_t)row[rt] * a.ldu + 32 + 8 * g + 4 * h) = o;
        }
    }

    // ---- q|k|v projection: wave w owns n-tiles 12 w .. 12 w + 11; stream order [nt][ks]; two tiles (32 fragments) per ring round.
    // A pair of tiles = 64 columns = one 128-byte line per row: staged in this wave's LDS area and stored row-major.
    const float* bs = reinterpret_cast<const float*>(smem + QW_BIAS);
    char* stg = smem + QW_STG + wave * (QW_ROWS * QW_YPITCH);
#pragma unroll
    for (int i0 = 0; i0 < 12; i0 += 2) {
        if (i0 + 2 < 12) { bnext[0] = bfrag(nt0 + i0 + 2); bnext[1] = bfrag(nt0 + i0 + 3); }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int i = i0 + hh;
            const int nt = nt0 + i;
            f32x16 acc[2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) acc[rt] = bf_bias_init(bs, nt, h);
#pragma unroll
            for (int ks = 0; ks < BF_KS; ++ks) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) acc[rt] = mfma32(ring[16 * hh + ks], yf[rt][ks], acc[rt]);
                if (i + 2 < 12) ring[16 * hh + ks] = nx[(16 * hh + ks) * 64];
            }
            const int t = nt / 16;                     // (wave-uniform: the tile's adapter)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                acc[rt] = mfma32(bext[hh], t == 0 ? hb[rt][0] : (t == 1 ? hb[rt][1] : hb[rt][2]), acc[rt]);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bf16x4 o = {(bf16_t)acc[rt][4 * g], (bf16_t)acc[rt][4 * g + 1], (bf16_t)acc[rt][4 * g + 2], (bf16_t)acc[rt][4 * g + 3]};
                    *reinterpret_cast<bf16x4*>(stg + (32 * rt + m) * QW_YPITCH + (32 * hh + 8 * g + 4 * h) * 2) = o;
                }
            }
        }
        nx += BF_RING * 64;
        bext[0] = bnext[0]; bext[1] = bnext[1];
        // the pair's 64 rows x 128 B, row-major: lane l of pass j = row 8 j + (l >> 3), 16-byte chunk l & 7
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = 8 * j + (lane >> 3), ch = lane & 7;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + r * QW_YPITCH + ch * 16);
            if (m0 + r < a.M) *reinterpret_cast<bf16x8*>(a.Y + (size_t)(m0 + r) * a.ldy + 32 * (nt0 + i0) + 8 * ch) = v;
        }
    }
    if (BF_TOUCH && qw_touch_fold(touched) == 0x7fc07fc1u && a.M < 0) a.mean[0] = 0.f;      // (keeps the prefetch loads alive; never true)
}

struct QwBwd {
    int M;
    const bf16_t* dY; int lddy;
    const bf16_t* dres;
    const bf16_t* x;
    const float* gamma; const float* mean; const float* rstd;
    const bf16x8* Wst;
    const bf16_t* At; int ldat;
    const bf16_t* Bbt; int ldbt;
    float alpha; float p; const long long* seed; unsigned sites[3];
    bf16_t* V; int ldv;
    bf16_t* dx;
};

// Stream per wave w (W_bwd_wide): Wqkv^T fragments of its feature tiles 2 w, 2 w + 1 in order [ks = 0 .. 95][c2] (192 fragments); a ring
// round = 16 k-steps = one staged chunk of dY.
template <bool DROP>
__global__ __launch_bounds__(256, 1) void block_qkv_wide_bwd_kernel(QwBwd a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * QW_ROWS;
    int row[2];
    bool rvalid[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { row[t] = min(m0 + 32 * t + m, a.M - 1); rvalid[t] = m0 + 32 * t + m < a.M; }

    // ---- requests: the first dY chunk, the ring, then x / dres tiles, LayerNorm operands, adapter operands, touches
    bf16x8 dv[8];
    qw_load_rows256(a.dY, a.lddy, 0, m0, a.M, dv);
    const bf16x8* nx = a.Wst + (size_t)wave * 192 * 64 + lane;
    bf16x8 ring[BF_RING];
#pragma unroll
    for (int i = 0; i < BF_RING; ++i) ring[i] = nx[i * 64];
    nx += BF_RING * 64;
    {
        bf16x8 xv[8];
        qw_load_rows256(a.x, BF_D, 0, m0, a.M, xv);
        qw_rows256_to_lds(smem + QW_XT, xv);           // (waits for dY chunk 0 and the ring as well: all are needed next anyway)
        if (a.dres != nullptr) {
            qw_load_rows256(a.dres, BF_D, 0, m0, a.M, xv);
            qw_rows256_to_lds(smem + QW_YT, xv);
        }
    }
    float mean[2], rstd[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { mean[t] = a.mean[row[t]]; rstd[t] = a.rstd[row[t]]; }
    if (wave == 0) reinterpret_cast<f32x4*>(smem + QW_PAR)[lane] = reinterpret_cast<const f32x4*>(a.gamma)[lane];
    const float* pgam = reinterpret_cast<const float*>(smem + QW_PAR);
    // A_t^T fragments of this wave's two feature tiles (chained k order over the adapter's 16 ranks)
    bf16x8 atf[2][3];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const bf16_t* ap = a.At + (size_t)(64 * wave + 32 * c2 + m) * a.ldat + 16 * t + 4 * h;
            const bf16x4 lo = *reinterpret_cast<const bf16x4*>(ap), hi = *reinterpret_cast<const bf16x4*>(ap + 8);
            atf[c2][t] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
    QwTouch touched;
    if (BF_TOUCH) touched = qw_touch_stream(a.Wst, 768);
    qw_rows256_to_lds(smem + QW_STG, dv);              // chunk 0 -> buffer 0
    __syncthreads();

    // B_blk^T fragment of k-step ks for V = s dY B_blk: rows 0..31 (q|k adapters, block-diagonal zeros select) or 32..47 (v)
    auto vfrag = [&](int ks) __attribute__((always_inline)) {
        const int vrow = (ks >= 64) ? 32 + (m & 15) : m;
        return *reinterpret_cast<const bf16x8*>(a.Bbt + (size_t)vrow * a.ldbt + 16 * ks + 8 * h);
    };
    // ---- dy^T[c, row] = sum_n Wqkv^T[c, n] dY^T[n, row] for this wave's 64 features c over all n, chunk by chunk; this wave's share
    // of V: the k-steps with (ks & 3) == wave
    // (no branch inside a chunk: every wave does its V steps at fixed points, on a B fragment it picks by address; the adapters
    // change class at k-step 64 = chunk 4: one accumulator, handed over once)
    f32x16 acc[2][2], v01[2], vacc[2] = {zero16(), zero16()};
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) { acc[c2][0] = zero16(); acc[c2][1] = zero16(); }
    bf16x8 vf = vfrag(wave);
#pragma unroll 1
    for (int q = 0; q < 6; ++q) {
        const char* buf = smem + QW_STG + (q & 1) * 32768;
        qw_load_rows256(a.dY, a.lddy, 256 * min(q + 1, 5), m0, a.M, dv);      // (the last round re-requests chunk 5: nobody reads it)
        bf16x8 bq[2][2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) bq[0][rt] = qw_frag256(buf + rt * 16384, m, h, 0);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k + 1 < 16) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) bq[(k + 1) & 1][rt] = qw_frag256(buf + rt * 16384, m, h, k + 1);
            }
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) acc[c2][rt] = mfma32(ring[2 * k + c2], bq[k & 1][rt], acc[c2][rt]);
                ring[2 * k + c2] = nx[(2 * k + c2) * 64];      // (the last round reads the 32 fragments behind this wave's stream)
            }
            if ((k & 3) == 3) {                        // this wave's V step of the group: k-step 16 q + (k - 3) + wave
                const int kl = k - 3 + wave;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) vacc[rt] = mfma32(vf, qw_frag256(buf + rt * 16384, m, h, kl), vacc[rt]);
                vf = vfrag(min(16 * q + kl + 4, 95));
            }
        }
        nx += BF_RING * 64;
        if (q == 3) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) { v01[rt] = vacc[rt]; vacc[rt] = zero16(); }
        }
        qw_rows256_to_lds(smem + QW_STG + ((q + 1) & 1) * 32768, dv);
        __syncthreads();
    }
    f32x16 (&vv)[2] = vacc;
    // ---- V partials (rows 0..31 of v01: q|k adapters; rows 0..15 of vv: v adapter) meet in LDS (the chunk buffers' place)
    {
        f32x4* vp = reinterpret_cast<f32x4*>(smem + QW_STG);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                vp[((wave * 4 + rt) * 4 + g) * 64 + lane] = f32x4{v01[rt][4 * g], v01[rt][4 * g + 1], v01[rt][4 * g + 2], v01[rt][4 * g + 3]};
                vp[((wave * 4 + 2 + rt) * 4 + g) * 64 + lane] = f32x4{vv[rt][4 * g], vv[rt][4 * g + 1], vv[rt][4 * g + 2], vv[rt][4 * g + 3]};
            }
        __syncthreads();
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 s0 = vp[((0 * 4 + rt) * 4 + g) * 64 + lane], s1 = vp[((0 * 4 + 2 + rt) * 4 + g) * 64 + lane];
#pragma unroll
                for (int w = 1; w < 4; ++w) { s0 += vp[((w * 4 + rt) * 4 + g) * 64 + lane]; s1 += vp[((w * 4 + 2 + rt) * 4 + g) * 64 + lane]; }
#pragma unroll
                for (int i = 0; i < 4; ++i) { v01[rt][4 * g + i] = s0[i] * a.alpha; vv[rt][4 * g + i] = s1[i] * a.alpha; }
            }
    }
    if (wave < 2 && rvalid[wave]) {
        const int rt = wave;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const bf16x4 o = {(bf16_t)v01[rt][4 * g], (bf16_t)v01[rt][4 * g + 1], (bf16_t)v01[rt][4 * g + 2], (bf16_t)v01[rt][4 * g + 3]};
            *reinterpret_cast<bf16x4*>(a.V + (size_t)row[rt] * a.ldv + 8 * g + 4 * h) = o;
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const bf16x4 o = {(bf16_t)vv[rt][4 * g], (bf16_t)vv[rt][4 * g + 1], (bf16_t)vv[rt][4 * g + 2], (bf16_t)vv[rt][4 * g + 3]};
            *reinterpret_cast<bf16x4*>(a.V + (size_t)row[rt] * a.ldv + 32 + 8 * g + 4 * h) = o;
        }
    }
    // ---- masked side term of this wave's features: dy += keep_t / (1-p) (A_t^T V_t^T), V as stored (bf16)
    {
        unsigned long long keys[3] = {0, 0, 0};
        unsigned thr = 0;
        if (DROP) {
#pragma unroll
            for (int t = 0; t < 3; ++t) keys[t] = cvft_drop_key(a.seed, a.sites[t]);
            thr = cvft_drop_thr(a.p);
        }
        const float inv_keep = DROP ? 1.f / (1.f - a.p) : 1.f;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            bf16x8 hbV[3];
#pragma unroll
            for (int i = 0; i < 8; ++i) { hbV[0][i] = (bf16_t)v01[rt][i]; hbV[1][i] = (bf16_t)v01[rt][8 + i]; hbV[2][i] = (bf16_t)vv[rt][i]; }
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const f32x16 side = mfma32(atf[c2][t], hbV[t], zero16());
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        bool kp[4] = {true, true, true, true};
                        if (DROP) cvft_keep4(keys[t], ((unsigned long long)row[rt] * BF_D + 64 * wave + 32 * c2 + 8 * g + 4 * h) >> 2, thr, kp);
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[c2][rt][4 * g + i] += kp[i] ? side[4 * g + i] * inv_keep : 0.f;
                    }
                }
        }
    }
    // ---- LayerNorm backward + residual branch: dx = dres + rstd (g.v - mean_c(g.v) - xhat mean_c(g.v.xhat)); x / dres from the tiles
    float sp[2][2], all[8][2];
    bf16x4 xr[2][2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        sp[0][rt] = 0.f; sp[1][rt] = 0.f;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 64 * wave + 32 * c2 + 8 * g + 4 * h;
                xr[rt][c2][g] = *reinterpret_cast<const bf16x4*>(smem + QW_XT + rt * 16384 + bf_tile_off(m, c));
                const f32x4 gg = *reinterpret_cast<const f32x4*>(pgam + c);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = 4 * g + i;
                    acc[c2][rt][e] *= gg[i];
                    sp[0][rt] += acc[c2][rt][e];
                    sp[1][rt] += acc[c2][rt][e] * (((float)xr[rt][c2][g][i] - mean[rt]) * rstd[rt]);
                }
            }
        sp[0][rt] += __shfl_xor(sp[0][rt], 32, 64);
        sp[1][rt] += __shfl_xor(sp[1][rt], 32, 64);
    }
    qw_exchange(smem, wave, lane, sp, all);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const float m1 = (all[0][rt] + all[1][rt] + all[2][rt] + all[3][rt]) * (1.f / BF_D);
        const float m2 = (all[4][rt] + all[5][rt] + all[6][rt] + all[7][rt]) * (1.f / BF_D);
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 64 * wave + 32 * c2 + 8 * g + 4 * h;
                bf16x4 dr = {0, 0, 0, 0};
                if (a.dres != nullptr) dr = *reinterpret_cast<const bf16x4*>(smem + QW_YT + rt * 16384 + bf_tile_off(m, c));
                bf16x4 dx;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = 4 * g + i;
                    dx[i] = (bf16_t)((float)dr[i] + rstd[rt] * (acc[c2][rt][e] - m1 - (((float)xr[rt][c2][g][i] - mean[rt]) * rstd[rt]) * m2));
                }
                *reinterpret_cast<bf16x4*>(smem + QW_XT + rt * 16384 + bf_tile_off(m, c)) = dx;      // (over this lane's own x values)
            }
    }
    __syncthreads();
    qw_store_rows256(smem + QW_XT, a.dx, m0, a.M);
    if (BF_TOUCH && qw_touch_fold(touched) == 0x7fc07fc1u && a.M < 0) a.dx[0] = (bf16_t)0.f;   // (keeps the prefetch loads alive; never true)
}

template <typename K>
static int qw_prepare(K kernel) {
    if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, QW_TOTAL) != hipSuccess) {
        cvft_set_error("block_qkv_wide: cannot reserve %d bytes of LDS", QW_TOTAL);
        return -2;
    }
    return 0;
}
template <bool DROP>
static int launch_qw_fwd(const QwFwd& a, hipStream_t st) {
    static int ready = 0;
    if (!ready) { if (qw_prepare(block_qkv_wide_fwd_kernel<DROP>)) return -2; ready = 1; }
    hipLaunchKernelGGL((block_qkv_wide_fwd_kernel<DROP>), dim3((a.M + QW_ROWS - 1) / QW_ROWS), dim3(256), QW_TOTAL, st, a);
    return 0;
}
template <bool DROP>
static int launch_qw_bwd(const QwBwd& a, hipStream_t st) {
    static int ready = 0;
    if (!ready) { if (qw_prepare(block_qkv_wide_bwd_kernel<DROP>)) return -2; ready = 1; }
    hipLaunchKernelGGL((block_qkv_wide_bwd_kernel<DROP>), dim3((a.M + QW_ROWS - 1) / QW_ROWS), dim3(256), QW_TOTAL, st, a);
    return 0;
}

// called by cvft_block_qkv_fwd / _bwd (block_qkv.hip) when args.wide != 0; arguments are already checked there
int block_qkv_wide_fwd_launch(const cvft_block_qkv_args* p, void* stream) {
    QwFwd a;
    a.M = p->M; a.x = (const bf16_t*)p->x; a.gamma = p->gamma; a.beta = p->beta; a.eps = p->eps; a.mean = p->mean; a.rstd = p->rstd;
    a.Wst = (const bf16x8*)p->W_fwd; a.bias = p->bias;
    a.A = (const bf16_t*)p->A; a.lda = p->lda; a.Bb = (const bf16_t*)p->Bb; a.ldb = p->ldb;
    a.alpha = p->alpha; a.p = p->p; a.seed = (const long long*)p->seed;
    for (int i = 0; i < 3; ++i) { a.sites[i] = p->sites[i]; a.xd[i] = (bf16_t*)p->xd[i]; }
    a.U = (bf16_t*)p->U; a.ldu = p->ldu; a.y_out = (bf16_t*)p->y_out; a.Y = (bf16_t*)p->Y; a.ldy = p->ldy;
    return p->p > 0.f ? launch_qw_fwd<true>(a, (hipStream_t)stream) : launch_qw_fwd<false>(a, (hipStream_t)stream);
}
int block_qkv_wide_bwd_launch(const cvft_block_qkv_bwd_args* p, void* stream) {
    QwBwd a;
    a.M = p->M; a.dY = (const bf16_t*)p->dY; a.lddy = p->lddy; a.dres = (const bf16_t*)p->dres; a.x = (const bf16_t*)p->x;
    a.gamma = p->gamma; a.mean = p->mean; a.rstd = p->rstd; a.Wst = (const bf16x8*)p->W_bwd;
    a.At = (const bf16_t*)p->At; a.ldat = p->ldat; a.Bbt = (const bf16_t*)p->Bbt; a.ldbt = p->ldbt;
    a.alpha = p->alpha; a.p = p->p; a.seed = (const long long*)p->seed;
    for (int i = 0; i < 3; ++i) a.sites[i] = p->sites[i];
    a.V = (bf16_t*)p->V; a.ldv = p->ldv; a.dx = (bf16_t*)p->dx;
    return p->p > 0.f ? launch_qw_bwd<true>(a, (hipStream_t)stream) : launch_qw_bwd<false>(a, (hipStream_t)stream);
}
