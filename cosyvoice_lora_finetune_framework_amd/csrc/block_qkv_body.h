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

struct QkvBwd {
    int M;
    const bf16_t* dY; int lddy;
    const bf16_t* dres;
    const bf16_t* x;
    const float* gamma; const float* mean; const float* rstd;
    const bf16x8* Wst; int wave_frags; int N3;
    const bf16_t* At; int ldat;
    const bf16_t* Bbt; int ldbt;
    float alpha; float p; const long long* seed; unsigned sites[3];
    bf16_t* V; int ldv;
    bf16_t* dx;
};

// The q|k|v head's backward from the loads to dx.  LINK (block_link_bwd_kernel, block_fused.hip: the PREVIOUS block's tail backward
// follows on the same rows): a.Wst is the linked stream (this wave's 192 fragments, then its tail fragments) and the ring is left
// on the tail's first 32 fragments; dx also stays in registers (dxo, as stored); the cold-weight touches go through LDS (no
// register to hold).
template <bool DROP, bool LINK>
__device__ __forceinline__ void bq_bwd_body(const QkvBwd& a, char* smem, bf16x8 (&ring)[BF_RING], const bf16x8*& nx, bf16x4 (&dxo)[2][4]) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * BF_ROWS;
    const int row = min(m0 + m, a.M - 1);
    const bool rvalid = m0 + m < a.M;
    constexpr int KSW = 24;                            // k-steps per wave: 3N / 16 / 4 (3N = 1536)
    const int ks0 = wave * KSW;

    // ---- loads in the order they are needed: this wave's quarter of dY (B fragments), LayerNorm operands, adapter operands, ring
    // (24 fragments = 96 registers: parked in LDS in fragment order -- wave-private, lane-linear, so no barrier and no bank
    // conflicts -- and read back one k-step at a time; the region is the partial-tile exchange area, free until the loop ends)
    bf16x8* dyl = reinterpret_cast<bf16x8*>(smem + BF_LDS_PART) + wave * KSW * 64 + lane;
#pragma unroll
    for (int k = 0; k < KSW; ++k) dyl[k * 64] = *reinterpret_cast<const bf16x8*>(a.dY + (size_t)row * a.lddy + 16 * (ks0 + k) + 8 * h);
    bf16x4 xr[2][4], dr[2][4];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 64 * wave + 32 * c2 + 8 * g + 4 * h;
            xr[c2][g] = *reinterpret_cast<const bf16x4*>(a.x + (size_t)row * BF_D + c);
            const bf16x4 z = {0, 0, 0, 0};
            dr[c2][g] = a.dres ? *reinterpret_cast<const bf16x4*>(a.dres + (size_t)row * BF_D + c) : z;
        }
    const float mean = a.mean[row], rstd = a.rstd[row];
    if (wave == 0) reinterpret_cast<f32x4*>(smem + BF_LDS_PAR)[lane] = reinterpret_cast<const f32x4*>(a.gamma)[lane];
    const float* pgam = reinterpret_cast<const float*>(smem + BF_LDS_PAR);
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
    // B_blk^T fragments for V = s dY B_blk: k-step ks (16 output features of adapter t = ks / 32) multiplies rows 16 t .. 16 t + 15;
    // one row tile per k-step: rows 0..31 (q|k adapters, block-diagonal zeros do the selection) or rows 32..47 (v; clamped)
    nx = a.Wst + (size_t)wave * a.wave_frags * 64 + lane;
#pragma unroll
    for (int i = 0; i < BF_RING; ++i) ring[i] = nx[i * 64];
    nx += BF_RING * 64;
    BfTouch touched;
    if (BF_TOUCH && !LINK) touched = bf_touch_stream(a.Wst, 4 * a.wave_frags);
    if (BF_TOUCH && LINK) bf_touch_stream_lds(a.Wst, 4 * a.wave_frags, smem + BF_LDS_DUMMY);
    constexpr int VD = 4;                              // V fragments requested ahead (k-steps)
    bf16x8 vf[VD];
    auto vfrag = [&](int k) {
        const int ks = ks0 + min(k, KSW - 1);
        const int vrow = (ks >= 64) ? 32 + (m & 15) : m;
        return *reinterpret_cast<const bf16x8*>(a.Bbt + (size_t)vrow * a.ldbt + 16 * ks + 8 * h);
    };
#pragma unroll
    for (int k = 0; k < VD; ++k) vf[k] = vfrag(k);

    // ---- dy^T[c, m] = sum_n Wqkv^T[c, n] dY^T[n, m] over this wave's n range; stream order [ks][ct], 4 k-steps per ring round
    f32x16 acc[BF_CT];
#pragma unroll
    for (int ct = 0; ct < BF_CT; ++ct) acc[ct] = zero16();
    f32x16 v01 = zero16(), vv = zero16();
#pragma unroll
    for (int r = 0; r < KSW / 4; ++r) {
        bf16x8 dyf[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) dyf[q] = dyl[(4 * r + q) * 64];
#pragma unroll
        for (int j = 0; j < BF_RING; ++j) {
            const int k = 4 * r + (j >> 3);
            acc[j & 7] = mfma32(ring[j], dyf[j >> 3], acc[j & 7]);
            if (r + 1 < KSW / 4) ring[j] = nx[j * 64];
            if ((j & 7) == 7) {                        // this k-step's share of V, and the fragment four k-steps ahead
                const bool third = (ks0 + k) >= 64;   // (wave-uniform: the v adapter's output features)
                if (third) vv = mfma32(vf[k % VD], dyf[j >> 3], vv); else v01 = mfma32(vf[k % VD], dyf[j >> 3], v01);
                vf[k % VD] = vfrag(k + VD);
            }
        }
        nx += BF_RING * 64;
    }
    // V partials (rows 0..31 of v01: q|k adapters; rows 0..15 of vv: v adapter) meet in LDS first, then the main term's partials
    __syncthreads();                                   // (every wave is done with its parked dY fragments: the area is reused)
    {
        f32x4* vp = reinterpret_cast<f32x4*>(smem + BF_LDS_PART);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            vp[((wave * 2 + 0) * 4 + g) * 64 + lane] = f32x4{v01[4 * g], v01[4 * g + 1], v01[4 * g + 2], v01[4 * g + 3]};
            vp[((wave * 2 + 1) * 4 + g) * 64 + lane] = f32x4{vv[4 * g], vv[4 * g + 1], vv[4 * g + 2], vv[4 * g + 3]};
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 s0 = vp[((0 * 2 + 0) * 4 + g) * 64 + lane], s1 = vp[((0 * 2 + 1) * 4 + g) * 64 + lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) { s0 += vp[((w * 2 + 0) * 4 + g) * 64 + lane]; s1 += vp[((w * 2 + 1) * 4 + g) * 64 + lane]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) { v01[4 * g + i] = s0[i] * a.alpha; vv[4 * g + i] = s1[i] * a.alpha; }
        }
        __syncthreads();                               // (every wave has read the V partials before `part` is rewritten)
    }
    float v[2][16];
    bf_reduce(smem, wave, lane, acc, v);               // v = dL/dy (main term) for this wave's 64 features
    if (LINK) {
        // the tail's first 32 fragments (positions 192..223 of the linked stream) are requested HERE: the accumulators are dead, the
        // ring's registers are free again, and the side term / LayerNorm backward below cover the latency
        nx -= BF_RING * 64;
#pragma unroll
        for (int i = 0; i < BF_RING; ++i) ring[i] = nx[i * 64];
        nx += BF_RING * 64;
    }
    if (wave == 0 && rvalid) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const bf16x4 o = {(bf16_t)v01[4 * g], (bf16_t)v01[4 * g + 1], (bf16_t)v01[4 * g + 2], (bf16_t)v01[4 * g + 3]};
            *reinterpret_cast<bf16x4*>(a.V + (size_t)row * a.ldv + 8 * g + 4 * h) = o;
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const bf16x4 o = {(bf16_t)vv[4 * g], (bf16_t)vv[4 * g + 1], (bf16_t)vv[4 * g + 2], (bf16_t)vv[4 * g + 3]};
            *reinterpret_cast<bf16x4*>(a.V + (size_t)row * a.ldv + 32 + 8 * g + 4 * h) = o;
        }
    }
    // ---- masked side term of this wave's features: v += keep_t / (1-p) (A_t^T V_t^T), V as stored (bf16)
    {
        bf16x8 hbV[3];
#pragma unroll
        for (int i = 0; i < 8; ++i) { hbV[0][i] = (bf16_t)v01[i]; hbV[1][i] = (bf16_t)v01[8 + i]; hbV[2][i] = (bf16_t)vv[i]; }
        unsigned long long keys[3] = {0, 0, 0};
        unsigned thr = 0;
        if (DROP) {
#pragma unroll
            for (int t = 0; t < 3; ++t) keys[t] = cvft_drop_key(a.seed, a.sites[t]);
            thr = cvft_drop_thr(a.p);
        }
        const float inv_keep = DROP ? 1.f / (1.f - a.p) : 1.f;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const f32x16 side = mfma32(atf[c2][t], hbV[t], zero16());
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bool kp[4] = {true, true, true, true};
                    if (DROP) cvft_keep4(keys[t], ((unsigned long long)row * BF_D + 64 * wave + 32 * c2 + 8 * g + 4 * h) >> 2, thr, kp);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[c2][4 * g + i] += kp[i] ? side[4 * g + i] * inv_keep : 0.f;
                }
            }
    }
    // ---- LayerNorm backward + residual branch: dx = dres + rstd (g.v - mean_c(g.v) - xhat mean_c(g.v.xhat))
    float xh[2][16], gv[2][16];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 gg = *reinterpret_cast<const f32x4*>(pgam + 64 * wave + 32 * c2 + 8 * g + 4 * h);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 4 * g + i;
                xh[c2][e] = ((float)xr[c2][g][i] - mean) * rstd;
                gv[c2][e] = gg[i] * v[c2][e];
                s1 += gv[c2][e];
                s2 += gv[c2][e] * xh[c2][e];
            }
        }
    const float m1 = bf_rowsum(smem, 0, wave, lane, s1) * (1.f / BF_D);
    const float m2 = bf_rowsum(smem, 1, wave, lane, s2) * (1.f / BF_D);
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 64 * wave + 32 * c2 + 8 * g + 4 * h;
            bf16x4 dx;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 4 * g + i;
                dx[i] = (bf16_t)((float)dr[c2][g][i] + rstd * (gv[c2][e] - m1 - xh[c2][e] * m2));
            }
            if (rvalid) *reinterpret_cast<bf16x4*>(a.dx + (size_t)row * BF_D + c) = dx;
            if (LINK) dxo[c2][g] = dx;
        }
    if (BF_TOUCH && !LINK && bf_touch_fold(touched) == 0x7fc07fc1u && a.M < 0) a.dx[0] = (bf16_t)0.f;   // (keeps the prefetch loads alive; never true)
}

