// block_qkv.hip -- first half of the estimator's transformer block (matcha transformer.py:255-289 == modules.py:349-361) as
// row-tile chain kernels (structure and helpers: block_fused.hip / block_common.h):
//
//   cvft_block_qkv_fwd:  y = norm1(x);  Y = y Wqkv^T + b + (s / (1-p) sum_t drop_t(y) A_t^T) B_blk^T           (one launch)
//                        -- LayerNorm, the three LoRA adapters' rank-16 side products under lora_dropout (lora.py:70-73) and the
//                        stacked q|k|v projection; y never touches HBM (only the dropped copies the adapter gradients need).
//   cvft_block_qkv_bwd:  V = s dY B_blk;  dy = dY Wqkv + sum_t keep_t / (1-p) (V_t A_t);  dx = dres + norm1'(dy)  (one launch)
//
// The q|k|v weight arrives as a per-wave stream of pre-packed MFMA A-operand fragments (ring of 32); the adapters change every
// optimiser step, so their operands are read straight from the optimiser's bf16 shadows (a few KB, fragment-shaped loads).
// Masks: the counter-based masks of common.h (same (seed, site, element index) -> same mask as cvft_skinny_dropout / the masked
// rank extension of cvft_gemm), so the launch-per-stage form and this one agree mask for mask.
#include "block_qkv_body.h"

template <bool DROP, int NS>
__global__ __launch_bounds__(256, 1) void block_qkv_fwd_kernel(QkvFwd a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * BF_ROWS;
    const int row = min(m0 + m, a.M - 1);
    const bool rvalid = m0 + m < a.M;
    const int half = NS > 1 ? blockIdx.y : 0;

    // ---- loads in the order they are needed: x, small parameters (-> LDS), adapter operands, then the ring
    bf16x4 xb[2][4];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int g = 0; g < 4; ++g) xb[c2][g] = *reinterpret_cast<const bf16x4*>(a.x + (size_t)row * BF_D + 64 * wave + 32 * c2 + 8 * g + 4 * h);
    {
        BqParams pr;
        bq_params_load(a, wave, lane, pr);
        bq_params_store(a, smem, wave, lane, pr);
    }
    const int nt0 = (48 / NS) * half + (12 / NS) * wave;     // this wave's first n-tile
    BqOps<NS> ops;
    bq_load_ops<NS>(a, wave, m, h, nt0, ops);
    // (the packed stream is [48 n-tiles][16 k-steps] in tile order: tile nt0 starts at fragment 16 nt0)
    const bf16x8* nx = a.Wst + (size_t)nt0 * BF_KS * 64 + lane;
    bf16x8 ring[BF_RING];
#pragma unroll
    for (int i = 0; i < BF_RING; ++i) ring[i] = nx[i * 64];
    nx += BF_RING * 64;
    BfTouch touched;
    if (BF_TOUCH) touched = bf_touch_stream(a.Wst, 4 * a.wave_frags);
    bq_head_fwd<DROP, NS>(a, smem, xb, ring, nx, ops, lane, wave, row, rvalid, nt0, half == 0);
    if (BF_TOUCH && bf_touch_fold(touched) == 0x7fc07fc1u && a.M < 0) a.mean[0] = 0.f;      // (keeps the prefetch loads alive; never true)
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

template <bool DROP>
__global__ __launch_bounds__(256, 1) void block_qkv_bwd_kernel(QkvBwd a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
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
    const bf16x8* nx = a.Wst + (size_t)wave * a.wave_frags * 64 + lane;
    bf16x8 ring[BF_RING];
#pragma unroll
    for (int i = 0; i < BF_RING; ++i) ring[i] = nx[i * 64];
    nx += BF_RING * 64;
    BfTouch touched;
    if (BF_TOUCH) touched = bf_touch_stream(a.Wst, 4 * a.wave_frags);
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
        }
    if (BF_TOUCH && bf_touch_fold(touched) == 0x7fc07fc1u && a.M < 0) a.dx[0] = (bf16_t)0.f;   // (keeps the prefetch loads alive; never true)
}

template <bool DROP, int NS>
static int launch_qkv_fwd(const QkvFwd& a, hipStream_t st) {
    static int ready = 0;
    if (!ready) { if (bf_prepare(block_qkv_fwd_kernel<DROP, NS>)) return -2; ready = 1; }
    hipLaunchKernelGGL((block_qkv_fwd_kernel<DROP, NS>), dim3((a.M + BF_ROWS - 1) / BF_ROWS, NS), dim3(256), BF_LDS_TOTAL, st, a);
    return 0;
}
template <bool DROP>
static int launch_qkv_bwd(const QkvBwd& a, hipStream_t st) {
    static int ready = 0;
    if (!ready) { if (bf_prepare(block_qkv_bwd_kernel<DROP>)) return -2; ready = 1; }
    hipLaunchKernelGGL((block_qkv_bwd_kernel<DROP>), dim3((a.M + BF_ROWS - 1) / BF_ROWS), dim3(256), BF_LDS_TOTAL, st, a);
    return 0;
}

int block_qkv_wide_fwd_launch(const cvft_block_qkv_args* p, void* stream);          // block_qkv_wide.hip
int block_qkv_wide_bwd_launch(const cvft_block_qkv_bwd_args* p, void* stream);

extern "C" int cvft_block_qkv_fwd(const cvft_block_qkv_args* p, void* stream) {
    CVFT_CHECK_ARG(p && p->M > 0 && p->N3 == 1536, "cvft_block_qkv_fwd: need M > 0 and 3N == 1536 (N3=%d)", p ? p->N3 : -1);
    CVFT_CHECK_ARG(p->x && p->gamma && p->beta && p->mean && p->rstd && p->W_fwd && p->A && p->Bb && p->U && p->Y, "cvft_block_qkv_fwd: null operand");
    CVFT_CHECK_ARG(cvft_drop_rate_ok(p->p) && (p->p == 0.f || p->seed), "cvft_block_qkv_fwd: bad dropout args (p == 0 or 2^-16 <= p <= 1 - 2^-16, seed)");
    CVFT_CHECK_ARG(al16(p->x) && al16(p->gamma) && al16(p->beta) && al16(p->W_fwd) && al16(p->A) && al16(p->Bb) && al16(p->Y) && (!p->bias || al16(p->bias)) &&
                   p->lda % 8 == 0 && p->ldb % 4 == 0 && p->ldu % 4 == 0 && p->ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(p->U) & 7) == 0 &&
                   (!p->y_out || al16(p->y_out)) && (!p->xd[0] || al16(p->xd[0])) && (!p->xd[1] || al16(p->xd[1])) && (!p->xd[2] || al16(p->xd[2])),
                   "cvft_block_qkv_fwd: operands must be 16-byte aligned (row pitches: lda %% 8, ldb / ldu / ldy %% 4)");
    if (p->wide) {
        CVFT_CHECK_ARG(p->ldy % 8 == 0, "cvft_block_qkv_fwd: the wide form stores Y in 16-byte pieces (ldy %% 8 == 0)");
        const int rc = block_qkv_wide_fwd_launch(p, stream);
        if (rc) return rc;
        CVFT_LAUNCH_CHECK("cvft_block_qkv_fwd (wide)");
        return 0;
    }
    QkvFwd a;
    a.M = p->M; a.x = (const bf16_t*)p->x; a.gamma = p->gamma; a.beta = p->beta; a.eps = p->eps; a.mean = p->mean; a.rstd = p->rstd;
    a.Wst = (const bf16x8*)p->W_fwd; a.wave_frags = p->N3 / 8; a.bias = p->bias; a.N3 = p->N3;
    a.A = (const bf16_t*)p->A; a.lda = p->lda; a.Bb = (const bf16_t*)p->Bb; a.ldb = p->ldb;
    a.alpha = p->alpha; a.p = p->p; a.seed = (const long long*)p->seed;
    for (int i = 0; i < 3; ++i) { a.sites[i] = p->sites[i]; a.xd[i] = (bf16_t*)p->xd[i]; }
    a.U = (bf16_t*)p->U; a.ldu = p->ldu; a.y_out = (bf16_t*)p->y_out; a.Y = (bf16_t*)p->Y; a.ldy = p->ldy;
    // two workgroups per row tile while the launch cannot fill the chip with one AND this chain is not competing with two others:
    // same-box A/B, flow_only (2 chains) 14.95 -> 14.71 ms, joint (3 chains) 23.37 -> 23.53 -- what shortens a chain on an idle chip
    // costs the step when the chip is shared.  CVFT_QKV_NSPLIT=1/2 forces.
    static const int ns_env = getenv("CVFT_QKV_NSPLIT") ? atoi(getenv("CVFT_QKV_NSPLIT")) : 0;
    const int ns = ns_env ? ns_env : (((p->M + BF_ROWS - 1) / BF_ROWS <= 128 && cvft_concurrent_chains() < 3) ? 2 : 1);
    hipStream_t st = (hipStream_t)stream;
    const int rc = p->p > 0.f ? (ns == 2 ? launch_qkv_fwd<true, 2>(a, st) : launch_qkv_fwd<true, 1>(a, st))
                              : (ns == 2 ? launch_qkv_fwd<false, 2>(a, st) : launch_qkv_fwd<false, 1>(a, st));
    if (rc) return rc;
    CVFT_LAUNCH_CHECK("cvft_block_qkv_fwd");
    return 0;
}

extern "C" int cvft_block_qkv_bwd(const cvft_block_qkv_bwd_args* p, void* stream) {
    CVFT_CHECK_ARG(p && p->M > 0 && p->N3 == 1536, "cvft_block_qkv_bwd: need M > 0 and 3N == 1536");
    CVFT_CHECK_ARG(p->dY && p->x && p->gamma && p->mean && p->rstd && p->W_bwd && p->At && p->Bbt && p->V && p->dx, "cvft_block_qkv_bwd: null operand");
    CVFT_CHECK_ARG(cvft_drop_rate_ok(p->p) && (p->p == 0.f || p->seed), "cvft_block_qkv_bwd: bad dropout args (p == 0 or 2^-16 <= p <= 1 - 2^-16, seed)");
    CVFT_CHECK_ARG(al16(p->dY) && al16(p->x) && al16(p->gamma) && al16(p->W_bwd) && al16(p->Bbt) && al16(p->dx) && (!p->dres || al16(p->dres)) &&
                   p->lddy % 8 == 0 && p->ldbt % 8 == 0 && p->ldat % 4 == 0 && p->ldv % 4 == 0 && (reinterpret_cast<uintptr_t>(p->At) & 7) == 0 &&
                   (reinterpret_cast<uintptr_t>(p->V) & 7) == 0, "cvft_block_qkv_bwd: operands must be 16-byte aligned (row pitches: lddy / ldbt %% 8, ldat / ldv %% 4)");
    if (p->wide) {
        const int rc = block_qkv_wide_bwd_launch(p, stream);
        if (rc) return rc;
        CVFT_LAUNCH_CHECK("cvft_block_qkv_bwd (wide)");
        return 0;
    }
    QkvBwd a;
    a.M = p->M; a.dY = (const bf16_t*)p->dY; a.lddy = p->lddy; a.dres = (const bf16_t*)p->dres; a.x = (const bf16_t*)p->x;
    a.gamma = p->gamma; a.mean = p->mean; a.rstd = p->rstd; a.Wst = (const bf16x8*)p->W_bwd; a.wave_frags = p->N3 / 8; a.N3 = p->N3;
    a.At = (const bf16_t*)p->At; a.ldat = p->ldat; a.Bbt = (const bf16_t*)p->Bbt; a.ldbt = p->ldbt;
    a.alpha = p->alpha; a.p = p->p; a.seed = (const long long*)p->seed;
    for (int i = 0; i < 3; ++i) a.sites[i] = p->sites[i];
    a.V = (bf16_t*)p->V; a.ldv = p->ldv; a.dx = (bf16_t*)p->dx;
    const int rc = p->p > 0.f ? launch_qkv_bwd<true>(a, (hipStream_t)stream) : launch_qkv_bwd<false>(a, (hipStream_t)stream);
    if (rc) return rc;
    CVFT_LAUNCH_CHECK("cvft_block_qkv_bwd");
    return 0;
}
