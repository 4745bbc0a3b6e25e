// block_qkv_body.h -- device body of the q|k|v head forward (norm1 + the three LoRA adapters' rank-side products + the stacked
// projection; block_qkv.hip has the description), in pieces, so that two kernels can run it: block_qkv_fwd_kernel (block_qkv.hip: the
// head alone, x from HBM) and block_link_fwd_kernel (block_fused.hip: behind the PREVIOUS block's tail on the same 32 rows, x in
// registers, the weight ring already running on the head's stream).
#pragma once
#include "block_common.h"

struct QkvFwd {
    int M;
    const bf16_t* x;
    const float* gamma; const float* beta; float eps;
    float* mean; float* rstd;
    const bf16x8* Wst; int wave_frags;
    const float* bias; int N3;
    const bf16_t* A; int lda;
    const bf16_t* Bb; int ldb;
    float alpha; float p; const long long* seed; unsigned sites[3];
    bf16_t* U; int ldu;
    bf16_t* xd[3];
    bf16_t* y_out;
    bf16_t* Y; int ldy;
};

// one lane's 8 consecutive elements of a [rows][256] tensor starting at element index e0 (a multiple of 8): the two mask groups
__device__ __forceinline__ bf16x8 bq_mask8(bf16x8 v, unsigned long long key, unsigned long long e0, unsigned thr) {
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

// the small parameters on their way to LDS (gamma -> PAR slot 1, beta -> PAR slot 2, bias -> BIAS): loaded into registers first, so
// that a kernel whose LDS still holds somebody else's parameters can request them early and store them behind its barrier
struct BqParams { f32x4 gb; f32x4 bias[2]; };
__device__ __forceinline__ void bq_params_load(const QkvFwd& a, int wave, int lane, BqParams& r) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    r.gb = z;
    if (wave == 1) r.gb = reinterpret_cast<const f32x4*>(a.gamma)[lane];
    if (wave == 2) r.gb = reinterpret_cast<const f32x4*>(a.beta)[lane];
#pragma unroll
    for (int i = 0; i < 2; ++i) {                      // N3 / 4 = 384 float4 over 256 threads
        const int k = (int)threadIdx.x + 256 * i;
        r.bias[i] = (a.bias != nullptr && k < a.N3 / 4) ? reinterpret_cast<const f32x4*>(a.bias)[k] : z;
    }
}
__device__ __forceinline__ void bq_params_store(const QkvFwd& a, char* smem, int wave, int lane, const BqParams& r) {
    f32x4* par = reinterpret_cast<f32x4*>(smem + BF_LDS_PAR);
    if (wave == 1) par[64 + lane] = r.gb;
    if (wave == 2) par[128 + lane] = r.gb;
    f32x4* bs = reinterpret_cast<f32x4*>(smem + BF_LDS_BIAS);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int k = (int)threadIdx.x + 256 * i;
        if (k < a.N3 / 4) bs[k] = r.bias[i];
    }
}

// adapter operands of one wave: A fragments of its four k-steps (rows 0..31 = q|k adapters, rows 32..47 = v adapter, clamped
// beyond) and the B_blk fragments of its n-tiles (chained k order: element j = rank 8 (j>>2) + 4 h + (j&3) of the tile's adapter)
template <int NS>
struct BqOps { bf16x8 af01[4], afv[4], bext[12 / NS]; };
template <int NS>
__device__ __forceinline__ void bq_load_ops(const QkvFwd& a, int wave, int m, int h, int nt0, BqOps<NS>& o) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ks = 4 * wave + k;
        o.af01[k] = *reinterpret_cast<const bf16x8*>(a.A + (size_t)m * a.lda + 16 * ks + 8 * h);
        o.afv[k] = *reinterpret_cast<const bf16x8*>(a.A + (size_t)(32 + (m & 15)) * a.lda + 16 * ks + 8 * h);
    }
#pragma unroll
    for (int i = 0; i < 12 / NS; ++i) {
        const int nt = nt0 + i;
        const int t = nt / 16;                         // tiles per adapter: (3N / 3) / 32
        const bf16_t* bp = a.Bb + (size_t)(32 * nt + m) * a.ldb + 16 * t + 4 * h;
        const bf16x4 lo = *reinterpret_cast<const bf16x4*>(bp), hi = *reinterpret_cast<const bf16x4*>(bp + 8);
        o.bext[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
}

// From the LayerNorm to the stores of Y.  On entry: xb = this wave's 64 features of the 32 rows (as stored: bf16); the parameters
// are on their way to LDS (bq_params_store done by every wave; the first barrier inside publishes them); ring = stream positions
// 0..31 of this wave's n-tiles, nx = position 32.
// DROP: lora_dropout masks on (p > 0); otherwise the adapters see y itself.
// NS (1 | 2): output-column split over gridDim.y workgroups per row tile -- the projection's columns are independent, so half
// (blockIdx.y) streams half of the weight (its waves' n-tiles 24 half + 6 wave + i of the same packed stream) and repeats the
// cheap part (LayerNorm, rank-side product); only half 0 writes mean / rstd / U / the dropped copies.  At M = 2000 a launch is
// 63 row tiles on 256 CUs and stream-bound per CU: two workgroups per tile halve the bytes per CU.
template <bool DROP, int NS>
__device__ __forceinline__ void bq_head_fwd(const QkvFwd& a, char* smem, const bf16x4 (&xb)[2][4], bf16x8 (&ring)[BF_RING], const bf16x8* nx,
                                            const BqOps<NS>& ops, int lane, int wave, int row, bool rvalid, int nt0, bool writer) {
    const int m = lane & 31, h = lane >> 5;
    constexpr int ntw = 12 / NS;                       // n-tiles per wave (3N = 1536)
    constexpr int tiles_per_adapter = 16;              // (3N / 3) / 32
    const float* pgam = reinterpret_cast<const float*>(smem + BF_LDS_PAR) + BF_D, *pbet = pgam + BF_D;

    // ---- LayerNorm -> y tile in LDS -> every wave's B fragments
    float s = 0.f;
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) s += (float)xb[c2][g][i];
    const float mean = bf_rowsum(smem, 0, wave, lane, s) * (1.f / BF_D);
    s = 0.f;
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float d = (float)xb[c2][g][i] - mean; s += d * d; }
    const float rstd = rsqrtf(bf_rowsum(smem, 1, wave, lane, s) * (1.f / BF_D) + a.eps);
    if (writer && wave == 0 && lane < 32 && rvalid) { a.mean[row] = mean; a.rstd[row] = rstd; }
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 64 * wave + 32 * c2 + 8 * g + 4 * h;
            const f32x4 gg = *reinterpret_cast<const f32x4*>(pgam + c);
            const f32x4 be = *reinterpret_cast<const f32x4*>(pbet + c);
            bf16x4 y;
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = (bf16_t)(((float)xb[c2][g][i] - mean) * rstd * gg[i] + be[i]);
            *reinterpret_cast<bf16x4*>(smem + BF_LDS_TILE + bf_tile_off(m, c)) = y;
        }
    __syncthreads();
    bf16x8 yf[BF_KS];
    bf_tile_read(smem, lane, yf);

    // ---- rank-side product: U^T[r, m] = sum_k A_t[r, k] drop_t(y)[m, k]; wave w takes k-steps 4w .. 4w+3, partials meet in LDS
    f32x16 u01 = zero16(), uv = zero16();
    {
        unsigned long long keys[3] = {0, 0, 0};
        unsigned thr = 0;
        if (DROP) {
#pragma unroll
            for (int t = 0; t < 3; ++t) keys[t] = cvft_drop_key(a.seed, a.sites[t]);
            thr = cvft_drop_thr(a.p);
        }
        const float inv_keep = DROP ? 1.f / (1.f - a.p) : 1.f;
        const bf16x8 zf = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ks = 4 * wave + k;               // (runtime index: re-read this k-step's fragment from the LDS tile)
            const bf16x8 yk = *reinterpret_cast<const bf16x8*>(smem + BF_LDS_TILE + m * 512 + (((2 * ks + h) ^ (m & 15)) << 4));
            const unsigned long long e0 = (unsigned long long)row * BF_D + 16 * ks + 8 * h;
            bf16x8 vm[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                vm[t] = DROP ? bq_mask8(yk, keys[t], e0, thr) : yk;
                if (DROP && writer && a.xd[t] != nullptr && rvalid) {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)vm[t][e] * inv_keep);
                    *reinterpret_cast<bf16x8*>(a.xd[t] + e0) = o;
                }
            }
            if (!DROP && writer && a.y_out != nullptr && rvalid) *reinterpret_cast<bf16x8*>(a.y_out + e0) = yk;
            const bf16x8 aq = m < 16 ? ops.af01[k] : zf, ak = m < 16 ? zf : ops.af01[k], av = m < 16 ? ops.afv[k] : zf;
            u01 = mfma32(aq, vm[0], u01);
            u01 = mfma32(ak, vm[1], u01);
            uv = mfma32(av, vm[2], uv);
        }
        f32x4* part = reinterpret_cast<f32x4*>(smem + BF_LDS_PART);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            part[((wave * 2 + 0) * 4 + g) * 64 + lane] = f32x4{u01[4 * g], u01[4 * g + 1], u01[4 * g + 2], u01[4 * g + 3]};
            part[((wave * 2 + 1) * 4 + g) * 64 + lane] = f32x4{uv[4 * g], uv[4 * g + 1], uv[4 * g + 2], uv[4 * g + 3]};
        }
        __syncthreads();
        const float sc = a.alpha * inv_keep;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 s0 = part[((0 * 2 + 0) * 4 + g) * 64 + lane], s1 = part[((0 * 2 + 1) * 4 + g) * 64 + lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) { s0 += part[((w * 2 + 0) * 4 + g) * 64 + lane]; s1 += part[((w * 2 + 1) * 4 + g) * 64 + lane]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) { u01[4 * g + i] = s0[i] * sc; uv[4 * g + i] = s1[i] * sc; }
        }
    }
    // U (bf16, as stored) is what the rank extension multiplies -- and what backward's dB = dY^T U reads
    bf16x8 hb01[2], hbv;
#pragma unroll
    for (int i = 0; i < 8; ++i) { hb01[0][i] = (bf16_t)u01[i]; hb01[1][i] = (bf16_t)u01[8 + i]; hbv[i] = (bf16_t)uv[i]; }
    if (writer && wave == 0 && rvalid) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const bf16x4 o = {(bf16_t)u01[4 * g], (bf16_t)u01[4 * g + 1], (bf16_t)u01[4 * g + 2], (bf16_t)u01[4 * g + 3]};
            *reinterpret_cast<bf16x4*>(a.U + (size_t)row * a.ldu + 8 * g + 4 * h) = o;
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const bf16x4 o = {(bf16_t)uv[4 * g], (bf16_t)uv[4 * g + 1], (bf16_t)uv[4 * g + 2], (bf16_t)uv[4 * g + 3]};
            *reinterpret_cast<bf16x4*>(a.U + (size_t)row * a.ldu + 32 + 8 * g + 4 * h) = o;
        }
    }

    // ---- q|k|v projection: wave w owns n-tiles [w ntw, (w+1) ntw); stream order [nt][ks]; two tiles (32 fragments) per ring round
    const float* bs = reinterpret_cast<const float*>(smem + BF_LDS_BIAS);
#pragma unroll
    for (int i0 = 0; i0 < ntw; i0 += 2) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int i = i0 + hh;
            const int nt = nt0 + i;
            f32x16 acc = bf_bias_init(bs, nt, h);
#pragma unroll
            for (int ks = 0; ks < BF_KS; ++ks) {
                acc = mfma32(ring[16 * hh + ks], yf[ks], acc);
                if (i + 2 < ntw) ring[16 * hh + ks] = nx[(16 * hh + ks) * 64];
            }
            const int t = nt / tiles_per_adapter;
            acc = mfma32(ops.bext[i], t == 0 ? hb01[0] : (t == 1 ? hb01[1] : hbv), acc);
            if (rvalid) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bf16x4 o = {(bf16_t)acc[4 * g], (bf16_t)acc[4 * g + 1], (bf16_t)acc[4 * g + 2], (bf16_t)acc[4 * g + 3]};
                    *reinterpret_cast<bf16x4*>(a.Y + (size_t)row * a.ldy + 32 * nt + 8 * g + 4 * h) = o;
                }
            }
        }
        nx += BF_RING * 64;
    }
}
