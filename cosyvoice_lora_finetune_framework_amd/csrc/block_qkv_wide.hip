// block_qkv_wide.hip -- the first half of the estimator block (block_qkv.hip: norm1 + LoRA q|k|v projection, and its backward) with
// 64 rows per workgroup: every streamed weight fragment feeds two MFMAs (one per 32-row tile), half the CU-time per row at the
// same bytes per workgroup (the tail's counterpart is block_wide.hip).  Same math, masks and operands as
// block_qkv.hip; what differs:
//   forward  -- x comes in and Y goes out as 16-byte-per-lane row-major accesses through LDS tiles (Y: 196 KB per workgroup, staged
//               per wave in pairs of n-tiles = whole 128-byte lines); the B fragments of LN(x) are read from its LDS tiles.
//   eight waves (two per SIMD, <= 256 registers each): wave w owns six n-tiles (forward) / feature tile w (backward);
//   backward -- wave w owns 32 output features of dy for both row tiles and walks ALL of 3N (block_qkv.hip splits 3N over the
//               waves and sums 128 accumulator registers per wave through LDS: twice that does not fit); dY is staged through a
//               double-buffered LDS tile in chunks of 256 columns (row-major loads one chunk ahead), so every dY fragment is read
//               from LDS by the four waves and fetched from memory once.  The weight stream is re-packed for that order
//               (blockpack.py, W_bwd_wide: wave w of 8 = its feature tile, [ks]).
#include "block_common.h"

#define QW_ROWS 64
#define QW_THREADS 512                          // 8 waves, two per SIMD: one wave per SIMD issues MFMAs at 54 % of the pipe's rate and every
                                                // other instruction of the same wave comes on top (tools/ub/mfma_valu.hip); with <= 256
                                                // registers per wave the row tiles' B fragments are read from LDS, one step ahead
#define QW_RING 16                              // weight fragments in flight per wave (8 x 16 KB per CU)
#define QW_AHEAD 2                              // B fragments requested this many steps ahead of their MFMAs (forward)
#define QW_AHEAD_BWD 1                          // (backward: 256 registers are full)
// LDS carve
#define QW_XT 0                                 // x tiles 2 x [32][256] swizzled (backward: x, then dx)
#define QW_YT 32768                             // forward: LN(x) tiles;  backward: dres tiles
#define QW_STG 65536                            // 72 KB: forward: partial sums of the rank-side product (64 KB), then per-wave Y staging (8 x 9 KB);
                                                //        backward: the two dY chunk buffers, then the V partial sums (64 KB)
#define QW_STAT (QW_STG + 73728)                // 2 quantities x 2 row tiles x 8 waves x 32 rows
#define QW_BIAS (QW_STAT + 4096)                // 3N floats
#define QW_PAR (QW_BIAS + 6144)                 // gamma | beta
#define QW_TOTAL (QW_PAR + 2048)                // 151 552 B
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
// row-major global [rows][ld] bf16, 256 columns from column c0 <-> the two swizzled [32][256] tiles: 4 chunks of 16 B per thread
__device__ __forceinline__ void qw_load_rows256(const bf16_t* src, size_t ld, int c0, int m0, int M, bf16x8 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = i * QW_THREADS + threadIdx.x;
        const int r = q >> 5, ch = q & 31;
        v[i] = *reinterpret_cast<const bf16x8*>(src + (size_t)min(m0 + r, M - 1) * ld + c0 + 8 * ch);
    }
}
__device__ __forceinline__ void qw_rows256_to_lds(char* tiles, const bf16x8 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = i * QW_THREADS + threadIdx.x;
        const int r = q >> 5, ch = q & 31;
        *reinterpret_cast<bf16x8*>(tiles + (r >> 5) * 16384 + (r & 31) * 512 + ((ch ^ (r & 15)) << 4)) = v[i];
    }
}
__device__ __forceinline__ void qw_store_rows256(const char* tiles, bf16_t* dst, int m0, int M) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = i * QW_THREADS + threadIdx.x;
        const int r = q >> 5, ch = q & 31;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(tiles + (r >> 5) * 16384 + (r & 31) * 512 + ((ch ^ (r & 15)) << 4));
        if (m0 + r < M) *reinterpret_cast<bf16x8*>(dst + (size_t)(m0 + r) * BF_D + 8 * ch) = v;
    }
}
// every wave publishes two per-row values per row tile; all[q * 8 + w][t] = wave w's value q
__device__ __forceinline__ void qw_exchange(char* smem, int wave, int lane, const float (&mine)[2][2], float (&all)[16][2]) {
    float* st = reinterpret_cast<float*>(smem + QW_STAT);
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int t = 0; t < 2; ++t)
            if (lane < 32) st[(q * 2 + t) * 256 + wave * 32 + lane] = mine[q][t];
    __syncthreads();
    const int m = lane & 31;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int w = 0; w < 8; ++w) all[q * 8 + w][t] = st[(q * 2 + t) * 256 + w * 32 + m];
}
struct QwTouch { unsigned v[2]; };
__device__ __forceinline__ QwTouch qw_touch_stream(const void* stream, int total_frags) {
    const int part = (blockIdx.x >> 3) & 7;
    const int lines = total_frags;                     // one 128-byte line in eight, an eighth of them per workgroup (768 fragments)
    const char* base = reinterpret_cast<const char*>(stream) + (size_t)part * lines * 128;
    QwTouch r;
#pragma unroll
    for (int i = 0; i < 2; ++i)
        r.v[i] = *reinterpret_cast<const unsigned*>(base + (size_t)min(i * QW_THREADS + (int)threadIdx.x, lines - 1) * 128);
    return r;                                          // (folded only at the kernel's end: no wait on these loads)
}
__device__ __forceinline__ unsigned qw_touch_fold(const QwTouch& r) { return r.v[0] | r.v[1]; }

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
__global__ __launch_bounds__(QW_THREADS, 1) void block_qkv_wide_fwd_kernel(QwFwd a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * QW_ROWS;
    int row[2];
    bool rvalid[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { row[t] = min(m0 + 32 * t + m, a.M - 1); rvalid[t] = m0 + 32 * t + m < a.M; }

    BF_STAMP(0);
    // ---- requests in the order their data is needed: x tiles, small parameters, adapter operands, the ring, touches
    bf16x8 xv[4];
    qw_load_rows256(a.x, BF_D, 0, m0, a.M, xv);
    f32x4 pv = {0.f, 0.f, 0.f, 0.f}, bv = {0.f, 0.f, 0.f, 0.f};
    if (wave < 2) pv = reinterpret_cast<const f32x4*>(wave == 0 ? a.gamma : a.beta)[lane];
    if ((int)threadIdx.x < 384 && a.bias != nullptr) bv = reinterpret_cast<const f32x4*>(a.bias)[threadIdx.x];
    // rank-side product: wave w takes row tile w & 1 and k-steps 4 (w >> 1) .. + 3; its adapter A operand (rows 0..31 = q|k adapters,
    // rows 32..47 = v adapter, clamped beyond)
    const int urt = wave & 1, uk0 = 4 * (wave >> 1);
    bf16x8 af01[4], afv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        af01[k] = *reinterpret_cast<const bf16x8*>(a.A + (size_t)m * a.lda + 16 * (uk0 + k) + 8 * h);
        afv[k] = *reinterpret_cast<const bf16x8*>(a.A + (size_t)(32 + (m & 15)) * a.lda + 16 * (uk0 + k) + 8 * h);
    }
    // B_blk fragment of n-tile nt (chained k order: element j = rank 8 (j>>2) + 4 h + (j&3) of the tile's adapter)
    const int nt0 = 6 * wave;                           // this wave's six n-tiles
    auto bfrag = [&](int nt) __attribute__((always_inline)) {
        const int t = nt / 16;
        const bf16_t* bp = a.Bb + (size_t)(32 * nt + m) * a.ldb + 16 * t + 4 * h;
        const bf16x4 lo = *reinterpret_cast<const bf16x4*>(bp), hi = *reinterpret_cast<const bf16x4*>(bp + 8);
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    bf16x8 bext = bfrag(nt0);
    const bf16x8* nx = a.Wst + (size_t)nt0 * BF_KS * 64 + lane;     // (the packed stream is [48 n-tiles][16 k-steps])
    bf16x8 ring[QW_RING];
#pragma unroll
    for (int i = 0; i < QW_RING; ++i) ring[i] = nx[i * 64];
    nx += QW_RING * 64;
    QwTouch touched;
    if (BF_TOUCH) touched = qw_touch_stream(a.Wst, 768);
    qw_rows256_to_lds(smem + QW_XT, xv);
    if (wave < 2) reinterpret_cast<f32x4*>(smem + QW_PAR)[wave * 64 + lane] = pv;
    if ((int)threadIdx.x < 384) reinterpret_cast<f32x4*>(smem + QW_BIAS)[threadIdx.x] = bv;
    const float* pgam = reinterpret_cast<const float*>(smem + QW_PAR), *pbet = pgam + BF_D;
    __syncthreads();
    BF_STAMP(1);

    // ---- LayerNorm: wave w normalises features 32 w .. 32 w + 31 of both row tiles (statistics with one exchange) -> y tiles
    {
        bf16x4 xb[2][4];
        float st[2][2], tot[16][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float sw = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                xb[t][g] = *reinterpret_cast<const bf16x4*>(smem + QW_XT + t * 16384 + bf_tile_off(m, 32 * wave + 8 * g + 4 * h));
#pragma unroll
                for (int i = 0; i < 4; ++i) sw += (float)xb[t][g][i];
            }
            sw += __shfl_xor(sw, 32, 64);
            const float mw = sw * (1.f / 32.f);
            float qw = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float d = (float)xb[t][g][i] - mw; qw += d * d; }
            qw += __shfl_xor(qw, 32, 64);
            st[0][t] = mw; st[1][t] = qw;
        }
        qw_exchange(smem, wave, lane, st, tot);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float mu = 0.f, m2 = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) { mu += tot[w][t]; m2 += tot[8 + w][t]; }
            mu *= 0.125f;
#pragma unroll
            for (int w = 0; w < 8; ++w) { const float d = tot[w][t] - mu; m2 = fmaf(32.f * d, d, m2); }
            const float rstd = rsqrtf(m2 * (1.f / BF_D) + a.eps);
            if (wave == 0 && lane < 32 && rvalid[t]) { a.mean[row[t]] = mu; a.rstd[row[t]] = rstd; }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 32 * wave + 8 * g + 4 * h;
                const f32x4 gg = *reinterpret_cast<const f32x4*>(pgam + c);
                const f32x4 be = *reinterpret_cast<const f32x4*>(pbet + c);
                bf16x4 y;
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] = (bf16_t)(((float)xb[t][g][i] - mu) * rstd * gg[i] + be[i]);
                *reinterpret_cast<bf16x4*>(smem + QW_YT + t * 16384 + bf_tile_off(m, c)) = y;
            }
        }
    }
    __syncthreads();
    BF_STAMP(2);

    // ---- rank-side product: U^T[r, row] = sum_k A_t[r, k] drop_t(y)[row, k] over this wave's four k-steps of its row tile; the four
    // partials of a row tile meet in LDS.  A B-fragment lane holds 8 consecutive elements of its row: the mask groups of e0.
    const float inv_keep = DROP ? 1.f / (1.f - a.p) : 1.f;
    bf16x8 hb[2][3];
    {
        f32x16 u01 = zero16(), uv = zero16();
        unsigned long long keys[3] = {0, 0, 0};
        unsigned thr = 0;
        if (DROP) {
#pragma unroll
            for (int t = 0; t < 3; ++t) keys[t] = cvft_drop_key(a.seed, a.sites[t]);
            thr = cvft_drop_thr(a.p);
        }
        const bf16x8 zf = {0, 0, 0, 0, 0, 0, 0, 0};
        const int urow = min(m0 + 32 * urt + m, a.M - 1);
        const bool uvalid = m0 + 32 * urt + m < a.M;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ks = uk0 + k;                    // (runtime index)
            const bf16x8 aq = m < 16 ? af01[k] : zf, ak = m < 16 ? zf : af01[k], av = m < 16 ? afv[k] : zf;
            const bf16x8 yk = qw_frag256(smem + QW_YT + urt * 16384, m, h, ks);
            const unsigned long long e0 = (unsigned long long)urow * BF_D + 16 * ks + 8 * h;
            bf16x8 vm[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                vm[t] = DROP ? qw_mask8(yk, keys[t], e0, thr) : yk;
                if (DROP && a.xd[t] != nullptr && uvalid) {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)vm[t][e] * inv_keep);
                    *reinterpret_cast<bf16x8*>(a.xd[t] + e0) = o;
                }
            }
            if (!DROP && a.y_out != nullptr && uvalid) *reinterpret_cast<bf16x8*>(a.y_out + e0) = yk;
            u01 = mfma32(aq, vm[0], u01);
            u01 = mfma32(ak, vm[1], u01);
            uv = mfma32(av, vm[2], uv);
        }
        BF_STAMP(3);
        f32x4* part = reinterpret_cast<f32x4*>(smem + QW_STG);         // [wave][2 tiles][4 g][64 lanes]
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            part[((wave * 2 + 0) * 4 + g) * 64 + lane] = f32x4{u01[4 * g], u01[4 * g + 1], u01[4 * g + 2], u01[4 * g + 3]};
            part[((wave * 2 + 1) * 4 + g) * 64 + lane] = f32x4{uv[4 * g], uv[4 * g + 1], uv[4 * g + 2], uv[4 * g + 3]};
        }
        __syncthreads();
        const float sc = a.alpha * inv_keep;
        // every wave needs U (bf16, as stored -- what backward's dB = dY^T U reads) of both row tiles as B fragments of the extension
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            float s01[16], sv[8];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 s0 = part[((rt * 2 + 0) * 4 + g) * 64 + lane], s1 = part[((rt * 2 + 1) * 4 + g) * 64 + lane];
#pragma unroll
                for (int w = 1; w < 4; ++w) { s0 += part[(((2 * w + rt) * 2 + 0) * 4 + g) * 64 + lane]; s1 += part[(((2 * w + rt) * 2 + 1) * 4 + g) * 64 + lane]; }
#pragma unroll
                for (int i = 0; i < 4; ++i) { s01[4 * g + i] = s0[i] * sc; if (g < 2) sv[4 * g + i] = s1[i] * sc; }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { hb[rt][0][i] = (bf16_t)s01[i]; hb[rt][1][i] = (bf16_t)s01[8 + i]; hb[rt][2][i] = (bf16_t)sv[i]; }
            if (wave == rt && rvalid[rt]) {            // (wave 0 stores row tile 0's U, wave 1 row tile 1's)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bf16x4 o = {(bf16_t)s01[4 * g], (bf16_t)s01[4 * g + 1], (bf16_t)s01[4 * g + 2], (bf16_t)s01[4 * g + 3]};
                    *reinterpret_cast<bf16x4*>(a.U + (size_t)row[rt] * a.ldu + 8 * g + 4 * h) = o;
                }
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const bf16x4 o = {(bf16_t)sv[4 * g], (bf16_t)sv[4 * g + 1], (bf16_t)sv[4 * g + 2], (bf16_t)sv[4 * g + 3]};
                    *reinterpret_cast<bf16x4*>(a.U + (size_t)row[rt] * a.ldu + 32 + 8 * g + 4 * h) = o;
                }
            }
        }
        __syncthreads();                               // (the partial sums are read: the area becomes the Y staging)
    }

    // ---- q|k|v projection: wave w owns n-tiles 6 w .. 6 w + 5; stream order [nt][ks], one tile per ring round; B fragments of the
    // y tiles from LDS, one step ahead.  A pair of tiles = 64 columns = one 128-byte line per row: staged in this wave's LDS area
    // and stored row-major.
    BF_STAMP(4);
    const float* bs = reinterpret_cast<const float*>(smem + QW_BIAS);
    char* stg = smem + QW_STG + wave * (QW_ROWS * QW_YPITCH);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        BF_STAMP(5 + 2 * i);
        const int nt = nt0 + i;
        const bf16x8 bnext = bfrag(min(nt + 1, 47));
        f32x16 acc[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) acc[rt] = bf_bias_init(bs, nt, h);
        bf16x8 bq[QW_AHEAD + 1][2];
#pragma unroll
        for (int ks = 0; ks < QW_AHEAD; ++ks)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) bq[ks][rt] = qw_frag256(smem + QW_YT + rt * 16384, m, h, ks);
#pragma unroll
        for (int ks = 0; ks < BF_KS; ++ks) {
            if (ks + QW_AHEAD < BF_KS) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) bq[(ks + QW_AHEAD) % (QW_AHEAD + 1)][rt] = qw_frag256(smem + QW_YT + rt * 16384, m, h, ks + QW_AHEAD);
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) acc[rt] = mfma32(ring[ks], bq[ks % (QW_AHEAD + 1)][rt], acc[rt]);
            if (i + 1 < 6) ring[ks] = nx[ks * 64];
            __builtin_amdgcn_sched_barrier(0);
        }
        nx += QW_RING * 64;
        BF_STAMP(6 + 2 * i);
        const int t = nt / 16;                         // (wave-uniform: the tile's adapter)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            acc[rt] = mfma32(bext, t == 0 ? hb[rt][0] : (t == 1 ? hb[rt][1] : hb[rt][2]), acc[rt]);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const bf16x4 o = {(bf16_t)acc[rt][4 * g], (bf16_t)acc[rt][4 * g + 1], (bf16_t)acc[rt][4 * g + 2], (bf16_t)acc[rt][4 * g + 3]};
                *reinterpret_cast<bf16x4*>(stg + (32 * rt + m) * QW_YPITCH + (32 * (i & 1) + 8 * g + 4 * h) * 2) = o;
            }
        }
        bext = bnext;
        if (i & 1) {
            // the pair's 64 rows x 128 B, row-major: lane l of pass j = row 8 j + (l >> 3), 16-byte chunk l & 7
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = 8 * j + (lane >> 3), ch = lane & 7;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + r * QW_YPITCH + ch * 16);
                if (m0 + r < a.M) *reinterpret_cast<bf16x8*>(a.Y + (size_t)(m0 + r) * a.ldy + 32 * (nt - 1) + 8 * ch) = v;
            }
        }
    }
    BF_STAMP(17);
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

// Stream per wave w of 8 (W_bwd_wide): Wqkv^T fragments of its feature tile w in order [ks = 0 .. 95] (96 fragments); a ring round =
// 16 k-steps = one staged chunk of dY.
template <bool DROP>
__global__ __launch_bounds__(QW_THREADS, 1) void block_qkv_wide_bwd_kernel(QwBwd a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * QW_ROWS;
    int row[2];
    bool rvalid[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { row[t] = min(m0 + 32 * t + m, a.M - 1); rvalid[t] = m0 + 32 * t + m < a.M; }

    BF_STAMP(20);
    // ---- requests: the first dY chunk, the ring, then x / dres tiles, LayerNorm operands, adapter operands, touches
    bf16x8 dv[4];
    qw_load_rows256(a.dY, a.lddy, 0, m0, a.M, dv);
    const bf16x8* nx = a.Wst + (size_t)wave * 96 * 64 + lane;
    bf16x8 ring[QW_RING];
#pragma unroll
    for (int i = 0; i < QW_RING; ++i) ring[i] = nx[i * 64];
    nx += QW_RING * 64;
    {
        bf16x8 xv[4];
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
    // A_t^T fragments of this wave's feature tile (chained k order over the adapter's 16 ranks)
    bf16x8 atf[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const bf16_t* ap = a.At + (size_t)(32 * wave + m) * a.ldat + 16 * t + 4 * h;
        const bf16x4 lo = *reinterpret_cast<const bf16x4*>(ap), hi = *reinterpret_cast<const bf16x4*>(ap + 8);
        atf[t] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
    QwTouch touched;
    if (BF_TOUCH) touched = qw_touch_stream(a.Wst, 768);
    qw_rows256_to_lds(smem + QW_STG, dv);              // chunk 0 -> buffer 0
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) { BF_PIN(mean[t]); BF_PIN(rstd[t]); }
    BF_STAMP(21);

    // B_blk^T fragment of k-step ks for V = s dY B_blk: rows 0..31 (q|k adapters, block-diagonal zeros select) or 32..47 (v)
    auto vfrag = [&](int ks) __attribute__((always_inline)) {
        const int vrow = (ks >= 64) ? 32 + (m & 15) : m;
        return *reinterpret_cast<const bf16x8*>(a.Bbt + (size_t)vrow * a.ldbt + 16 * ks + 8 * h);
    };
    // ---- dy^T[c, row] = sum_n Wqkv^T[c, n] dY^T[n, row] for this wave's 32 features c over all n, chunk by chunk; this wave's share
    // of V: row tile w & 1, the k-steps with (ks & 3) == (w >> 1).  No branch inside a chunk: the V steps sit at fixed points, on a
    // B fragment picked by address; the adapters change class at k-step 64 = chunk 4: one accumulator, handed over once.
    const int vrt = wave & 1, vk = wave >> 1;
    f32x16 acc[2] = {zero16(), zero16()}, v01 = zero16(), vacc = zero16();
#pragma unroll 1
    for (int q = 0; q < 6; ++q) {
        BF_STAMP(22 + q);
        const char* buf = smem + QW_STG + (q & 1) * 32768;
        // this chunk's four B_blk^T fragments and the next dY chunk are requested BEFORE the chunk's ring re-requests: vmcnt retires in
        // order, and a fragment needed in a few steps must not queue behind fragments needed a chunk later
        bf16x8 vfa[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) vfa[j] = vfrag(16 * q + 4 * j + vk);
        qw_load_rows256(a.dY, a.lddy, 256 * min(q + 1, 5), m0, a.M, dv);      // (the last round re-requests chunk 5: nobody reads it)
        __builtin_amdgcn_sched_barrier(0);             // (at 256 registers the compiler otherwise sinks these requests to their uses)
        bf16x8 bq[QW_AHEAD_BWD + 1][2];
#pragma unroll
        for (int k = 0; k < QW_AHEAD_BWD; ++k)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) bq[k][rt] = qw_frag256(buf + rt * 16384, m, h, k);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k + QW_AHEAD_BWD < 16) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) bq[(k + QW_AHEAD_BWD) % (QW_AHEAD_BWD + 1)][rt] = qw_frag256(buf + rt * 16384, m, h, k + QW_AHEAD_BWD);
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) acc[rt] = mfma32(ring[k], bq[k % (QW_AHEAD_BWD + 1)][rt], acc[rt]);
            ring[k] = nx[k * 64];                      // (the last round reads the 16 fragments behind this wave's stream)
            if ((k & 3) == 3)                          // this wave's V step of the group: k-step 16 q + (k - 3) + vk, row tile vrt
                vacc = mfma32(vfa[k >> 2], qw_frag256(buf + vrt * 16384, m, h, k - 3 + vk), vacc);
        }
        nx += QW_RING * 64;
        if (q == 3) { v01 = vacc; vacc = zero16(); }
        qw_rows256_to_lds(smem + QW_STG + ((q + 1) & 1) * 32768, dv);
        __syncthreads();
    }
    BF_STAMP(28);
    // ---- V partials (rows 0..31 of v01: q|k adapters; rows 0..15 of vacc: v adapter) meet in LDS (the chunk buffers' place); every
    // wave needs V (bf16, as stored) of both row tiles for the side term
    bf16x8 hbV[2][3];
    {
        f32x4* vp = reinterpret_cast<f32x4*>(smem + QW_STG);          // [wave][2 tiles][4 g][64 lanes]
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            vp[((wave * 2 + 0) * 4 + g) * 64 + lane] = f32x4{v01[4 * g], v01[4 * g + 1], v01[4 * g + 2], v01[4 * g + 3]};
            vp[((wave * 2 + 1) * 4 + g) * 64 + lane] = f32x4{vacc[4 * g], vacc[4 * g + 1], vacc[4 * g + 2], vacc[4 * g + 3]};
        }
        __syncthreads();
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            float s01[16], sv[8];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 s0 = vp[((rt * 2 + 0) * 4 + g) * 64 + lane], s1 = vp[((rt * 2 + 1) * 4 + g) * 64 + lane];
#pragma unroll
                for (int w = 1; w < 4; ++w) { s0 += vp[(((2 * w + rt) * 2 + 0) * 4 + g) * 64 + lane]; s1 += vp[(((2 * w + rt) * 2 + 1) * 4 + g) * 64 + lane]; }
#pragma unroll
                for (int i = 0; i < 4; ++i) { s01[4 * g + i] = s0[i] * a.alpha; if (g < 2) sv[4 * g + i] = s1[i] * a.alpha; }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { hbV[rt][0][i] = (bf16_t)s01[i]; hbV[rt][1][i] = (bf16_t)s01[8 + i]; hbV[rt][2][i] = (bf16_t)sv[i]; }
            if (wave == rt && rvalid[rt]) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bf16x4 o = {(bf16_t)s01[4 * g], (bf16_t)s01[4 * g + 1], (bf16_t)s01[4 * g + 2], (bf16_t)s01[4 * g + 3]};
                    *reinterpret_cast<bf16x4*>(a.V + (size_t)row[rt] * a.ldv + 8 * g + 4 * h) = o;
                }
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const bf16x4 o = {(bf16_t)sv[4 * g], (bf16_t)sv[4 * g + 1], (bf16_t)sv[4 * g + 2], (bf16_t)sv[4 * g + 3]};
                    *reinterpret_cast<bf16x4*>(a.V + (size_t)row[rt] * a.ldv + 32 + 8 * g + 4 * h) = o;
                }
            }
        }
    }
    BF_STAMP(29);
    // ---- masked side term of this wave's features: dy += keep_t / (1-p) (A_t^T V_t^T)
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
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const f32x16 side = mfma32(atf[t], hbV[rt][t], zero16());
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bool kp[4] = {true, true, true, true};
                    if (DROP) cvft_keep4(keys[t], ((unsigned long long)row[rt] * BF_D + 32 * wave + 8 * g + 4 * h) >> 2, thr, kp);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[rt][4 * g + i] += kp[i] ? side[4 * g + i] * inv_keep : 0.f;
                }
            }
    }
    BF_STAMP(30);
    // ---- LayerNorm backward + residual branch: dx = dres + rstd (g.v - mean_c(g.v) - xhat mean_c(g.v.xhat)); x / dres from the tiles
    float sp[2][2], all[16][2];
    bf16x4 xr[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        sp[0][rt] = 0.f; sp[1][rt] = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 32 * wave + 8 * g + 4 * h;
            xr[rt][g] = *reinterpret_cast<const bf16x4*>(smem + QW_XT + rt * 16384 + bf_tile_off(m, c));
            const f32x4 gg = *reinterpret_cast<const f32x4*>(pgam + c);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 4 * g + i;
                acc[rt][e] *= gg[i];
                sp[0][rt] += acc[rt][e];
                sp[1][rt] += acc[rt][e] * (((float)xr[rt][g][i] - mean[rt]) * rstd[rt]);
            }
        }
        sp[0][rt] += __shfl_xor(sp[0][rt], 32, 64);
        sp[1][rt] += __shfl_xor(sp[1][rt], 32, 64);
    }
    BF_STAMP(12);
    qw_exchange(smem, wave, lane, sp, all);
    BF_STAMP(13);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) { m1 += all[w][rt]; m2 += all[8 + w][rt]; }
        m1 *= 1.f / BF_D; m2 *= 1.f / BF_D;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 32 * wave + 8 * g + 4 * h;
            bf16x4 dr = {0, 0, 0, 0};
            if (a.dres != nullptr) dr = *reinterpret_cast<const bf16x4*>(smem + QW_YT + rt * 16384 + bf_tile_off(m, c));
            bf16x4 dx;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 4 * g + i;
                dx[i] = (bf16_t)((float)dr[i] + rstd[rt] * (acc[rt][e] - m1 - (((float)xr[rt][g][i] - mean[rt]) * rstd[rt]) * m2));
            }
            *reinterpret_cast<bf16x4*>(smem + QW_XT + rt * 16384 + bf_tile_off(m, c)) = dx;      // (over this lane's own x values)
        }
    }
    BF_STAMP(14);
    __syncthreads();
    BF_STAMP(15);
    qw_store_rows256(smem + QW_XT, a.dx, m0, a.M);
    BF_STAMP(31);
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
    hipLaunchKernelGGL((block_qkv_wide_fwd_kernel<DROP>), dim3((a.M + QW_ROWS - 1) / QW_ROWS), dim3(QW_THREADS), QW_TOTAL, st, a);
    return 0;
}
template <bool DROP>
static int launch_qw_bwd(const QwBwd& a, hipStream_t st) {
    static int ready = 0;
    if (!ready) { if (qw_prepare(block_qkv_wide_bwd_kernel<DROP>)) return -2; ready = 1; }
    hipLaunchKernelGGL((block_qkv_wide_bwd_kernel<DROP>), dim3((a.M + QW_ROWS - 1) / QW_ROWS), dim3(QW_THREADS), QW_TOTAL, st, a);
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
