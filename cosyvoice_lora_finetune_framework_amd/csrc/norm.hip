// norm.hip -- LayerNorm (one wavefront per row) and channel-last GroupNorm+Mish.
// HBM-bound wavefront-level ops; statistics in fp32, two-pass (mean, then centred
// variance) to match torch's numerics.
//
// Replaces (reference): nn.LayerNorm call sites in encoder_layer.py:90-106,
// subsampling.py:69-113/338-383, matcha transformer.py:255-316; Block1D /
// ResnetBlock1D GroupNorm(8)+Mish (modules.py:60-94) and InterpolateRegulator's
// GroupNorm(1)+Mish (length_regulator.py:34-41).
#include <stdlib.h>
#include "common.h"

// One wavefront per row.  VP (C % VEC == 0, C <= 64*VEC*NCH, aligned): the row is read ONCE with 16-byte loads
// into registers (NCH chunks per lane) and every later pass runs on registers; otherwise strided scalar passes.
template <typename T, bool VP>
__global__ void __launch_bounds__(256) ln_fwd_kernel(int rows, int C, const T* __restrict__ x,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float eps, int relu, float post, T* __restrict__ y,
                                                      float* __restrict__ mean, float* __restrict__ rstd) {
    constexpr int VEC = 16 / sizeof(T);
    constexpr int NCH = 4;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + (size_t)row * C;
    T* yr = y + (size_t)row * C;
    if (VP) {
        const int nch = C / VEC;
        float v[NCH][VEC];
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int ch = lane + q * 64;
            if (ch < nch) {
                uint4 raw = *reinterpret_cast<const uint4*>(xr + ch * VEC);
                const T* e = reinterpret_cast<const T*>(&raw);
#pragma unroll
                for (int k = 0; k < VEC; ++k) { v[q][k] = to_f32(e[k]); s += v[q][k]; }
            }
        }
        const float mu = wave_sum(s) / (float)C;
        float s2 = 0.f;
#pragma unroll
        for (int q = 0; q < NCH; ++q)
            if (lane + q * 64 < nch)
#pragma unroll
                for (int k = 0; k < VEC; ++k) { float d = v[q][k] - mu; s2 += d * d; }
        const float rs = 1.0f / sqrtf(wave_sum(s2) / (float)C + eps);
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int ch = lane + q * 64;
            if (ch < nch) {
                T o[VEC];
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    float t = (v[q][k] - mu) * rs * gamma[ch * VEC + k] + beta[ch * VEC + k];
                    if (relu) t = fmaxf(t, 0.f);
                    o[k] = from_f32<T>(t * post);
                }
                *reinterpret_cast<uint4*>(yr + ch * VEC) = *reinterpret_cast<uint4*>(o);
            }
        }
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
        return;
    }
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += to_f32(xr[c]);
    const float mu = wave_sum(s) / (float)C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) {
        float d = to_f32(xr[c]) - mu;
        v += d * d;
    }
    const float rs = 1.0f / sqrtf(wave_sum(v) / (float)C + eps);
    for (int c = lane; c < C; c += 64) {
        float o = (to_f32(xr[c]) - mu) * rs * gamma[c] + beta[c];
        if (relu) o = fmaxf(o, 0.f);
        yr[c] = from_f32<T>(o * post);
    }
    if (lane == 0) {
        mean[row] = mu;
        rstd[row] = rs;
    }
}

// MK (vector path only): also writes dxm = keep(site) / (1 - p) * dx, the gradient that the dropout in front of x's producer
// (x = residual + dropout(linear(.)), encoder_layer.py:95 / 104) hands to that linear -- its backward then needs no pass of its
// own over dx (cvft_layernorm_bwd_mask).  The mask is cvft_dropout_add's: flat element index row * C + c, groups of 4.
// SD (with MK): the linear in front carries a rank-16 adapter -- its backward starts with V = s * dxm B ([rows][16], lora.py:71-76),
// a latency-bound launch of its own on the backward chain.  The wave that writes a row of dxm also leaves it in LDS (somrow) for
// ln_bwd_side_kernel's matrix-core product.
template <typename T, bool VP, bool MK = false, bool SD = false, int NCH = 4, bool RLT = true>
__device__ __forceinline__ void ln_bwd_row(const int row, int C, const T* __restrict__ x,
                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                           int relu, float post, const T* __restrict__ dy,
                                           const T* __restrict__ dres, T* __restrict__ dx,
                                           float mp = 0.f, const long long* __restrict__ mseed = nullptr,
                                           unsigned msite = 0, T* __restrict__ dxm = nullptr, T* somrow = nullptr) {
    constexpr int VEC = 16 / sizeof(T);
    const int lane = threadIdx.x & 63;
    const T* xr = x + (size_t)row * C;
    const T* dr = dy + (size_t)row * C;
    T* ox = dx + (size_t)row * C;
    const T* rr = dres ? dres + (size_t)row * C : nullptr;      // second gradient branch of x, added into dx
    const float mu = mean[row], rs = rstd[row];
    if (VP) {
        const int nch = C / VEC;
        float xh[NCH][VEC], g[NCH][VEC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int ch = lane + q * 64;
            if (ch < nch) {
                uint4 rx = *reinterpret_cast<const uint4*>(xr + ch * VEC);
                uint4 rd = *reinterpret_cast<const uint4*>(dr + ch * VEC);
                const T* ex = reinterpret_cast<const T*>(&rx);
                const T* ed = reinterpret_cast<const T*>(&rd);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const int c = ch * VEC + k;
                    float h = (to_f32(ex[k]) - mu) * rs;
                    float gg = to_f32(ed[k]) * post;
                    // (gamma / beta fetched unconditionally: behind the run-time relu test they were one load and one branch per element)
                    const float gmv = gamma[c], btv = RLT ? beta[c] : 0.f;
                    if (RLT) gg = ((relu != 0) & ((h * gmv + btv) <= 0.f)) ? 0.f : gg;
                    gg *= gmv;
                    xh[q][k] = h;
                    g[q][k] = gg;
                    s1 += gg;
                    s2 += gg * h;
                }
            }
        }
        s1 = wave_sum(s1) / (float)C;
        s2 = wave_sum(s2) / (float)C;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int ch = lane + q * 64;
            if (ch < nch) {
                T o[VEC];
                float add[VEC];
#pragma unroll
                for (int k = 0; k < VEC; ++k) add[k] = 0.f;
                if (rr) {
                    uint4 ra = *reinterpret_cast<const uint4*>(rr + ch * VEC);
                    const T* ea = reinterpret_cast<const T*>(&ra);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) add[k] = to_f32(ea[k]);
                }
#pragma unroll
                for (int k = 0; k < VEC; ++k) o[k] = from_f32<T>(rs * (g[q][k] - s1 - xh[q][k] * s2) + add[k]);
                *reinterpret_cast<uint4*>(ox + ch * VEC) = *reinterpret_cast<uint4*>(o);
                if (MK) {
                    const unsigned long long key = cvft_drop_key(mseed, msite);
                    const unsigned thr = cvft_drop_thr(mp);
                    const float inv = 1.f / (1.f - mp);
                    const unsigned long long g0 = ((unsigned long long)row * C + (unsigned long long)ch * VEC) >> 2;   // C % VEC == 0
                    T om[VEC];
#pragma unroll
                    for (int gq = 0; gq < VEC / 4; ++gq) {
                        bool kp[4];
                        cvft_keep4(key, g0 + gq, thr, kp);
#pragma unroll
                        for (int e = 0; e < 4; ++e) om[4 * gq + e] = kp[e] ? from_f32<T>(to_f32(o[4 * gq + e]) * inv) : from_f32<T>(0.f);
                    }
                    *reinterpret_cast<uint4*>(dxm + (size_t)row * C + ch * VEC) = *reinterpret_cast<uint4*>(om);
                    if constexpr (SD) *reinterpret_cast<uint4*>(somrow + ch * VEC) = *reinterpret_cast<uint4*>(om);      // the row, for the side product
                }
            }
        }
        return;
    }
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < C; c += 64) {
        float xh = (to_f32(xr[c]) - mu) * rs;
        float g = to_f32(dr[c]) * post;
        if (relu && (xh * gamma[c] + beta[c]) <= 0.f) g = 0.f;
        g *= gamma[c];
        s1 += g;
        s2 += g * xh;
    }
    s1 = wave_sum(s1) / (float)C;
    s2 = wave_sum(s2) / (float)C;
    for (int c = lane; c < C; c += 64) {
        float xh = (to_f32(xr[c]) - mu) * rs;
        float g = to_f32(dr[c]) * post;
        if (relu && (xh * gamma[c] + beta[c]) <= 0.f) g = 0.f;
        g *= gamma[c];
        ox[c] = from_f32<T>(rs * (g - s1 - xh * s2) + (rr ? to_f32(rr[c]) : 0.f));
    }
}

// NCH = 16-byte chunks per lane the vector path is unrolled for (2 cover C <= 128 chunks -- the 1024-wide LLM rows -- with half the
// registers of 4: 11.4 -> ~9.7 us at 2 664 x 1 024 on the mask form, tools/bench_ln_side.py)
// RL = false: the caller passes relu == 0 (every LayerNorm of the two models but the length regulator's); as a compile-time fact
// it removes the beta loads and the y <= 0 test from the per-element work (11.3 -> 9.7 us at 2 664 x 1 024)
template <typename T, bool VP, bool MK = false, int NCH = 4, bool RL = true>
__global__ void __launch_bounds__(256) ln_bwd_kernel(int rows, int C, const T* __restrict__ x,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      int relu, float post, const T* __restrict__ dy,
                                                      const T* __restrict__ dres, T* __restrict__ dx,
                                                      float mp = 0.f, const long long* __restrict__ mseed = nullptr,
                                                      unsigned msite = 0, T* __restrict__ dxm = nullptr) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    ln_bwd_row<T, VP, MK, false, NCH, RL>(row, C, x, gamma, beta, mean, rstd, RL ? relu : 0, post, dy, dres, dx, mp, mseed, msite, dxm);
}

// The side-product form.  B^T ([16][C] bf16) is staged in LDS once per workgroup (read straight from L2 it was 32 KB per ROW: 170 MB
// per launch at 5 328 rows, +7 us); the workgroup walks groups of four rows (one per wave).  After the four waves have left their
// masked rows in LDS, V[4][16] = rows . B^T is ONE 16x16x32 MFMA chain per wave over a quarter of C (rows 4..15 of the A operand
// repeat rows 0..3 and are ignored), the four partial tiles meet in LDS.  (Each wave forming its own row's 16 dot products with
// v_dot2_f32_bf16 read all of B^T from LDS per row: 2.5 / 5 us more than the plain kernel at 2 664 / 5 328 rows.)  Rows are padded
// by 8 elements so the 16 rows of a B fragment read fall on different banks.
template <int NCH>
__global__ void __launch_bounds__(256) ln_bwd_side_kernel(int rows, int C, const bf16_t* __restrict__ x,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const bf16_t* __restrict__ dy, const bf16_t* __restrict__ dres,
                                                           bf16_t* __restrict__ dx, float mp, const long long* __restrict__ mseed,
                                                           unsigned msite, bf16_t* __restrict__ dxm, const bf16_t* __restrict__ sB,
                                                           float salpha, bf16_t* __restrict__ sV) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ln_smem[];
    const int P = C + 8;                                           // LDS row pitch (elements)
    bf16_t* Bl = reinterpret_cast<bf16_t*>(ln_smem);               // [16][P]
    bf16_t* Om = Bl + 16 * P;                                      // [4][P]: this pass's masked rows
    float* Pt = reinterpret_cast<float*>(Om + 4 * P);              // [4 waves][4 rows][16]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, kg = lane >> 4;
    for (int i = tid; i < 2 * C; i += 256) {                       // 16 * C / 8 chunks
        const int r = i / (C / 8), c = i % (C / 8);
        *reinterpret_cast<uint4*>(Bl + r * P + c * 8) = reinterpret_cast<const uint4*>(sB)[i];
    }
    __syncthreads();
    const int groups = (rows + 3) >> 2, kq = C >> 2;               // k columns per wave
    for (int g = blockIdx.x; g < groups; g += gridDim.x) {          // (workgroup-uniform: every wave meets every barrier)
        const int row = g * 4 + w;
        if (row < rows)
            ln_bwd_row<bf16_t, true, true, true, NCH, false>(row, C, x, gamma, beta, mean, rstd, 0, 1.f, dy, dres, dx, mp, mseed, msite, dxm,
                                                             Om + w * P);
        __syncthreads();
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = w * kq; k0 < (w + 1) * kq; k0 += 32) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(Om + (l15 & 3) * P + k0 + kg * 8);
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(Bl + l15 * P + k0 + kg * 8);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
        }
        if (kg == 0) {                                             // D[m = 4 kg + i][n = l15]: rows 0..3
#pragma unroll
            for (int i = 0; i < 4; ++i) Pt[(w * 4 + i) * 16 + l15] = acc[i];
        }
        __syncthreads();
        if (tid < 64) {
            const int r = tid >> 4, j = tid & 15;
            const float v = (Pt[(0 * 4 + r) * 16 + j] + Pt[(1 * 4 + r) * 16 + j]) + (Pt[(2 * 4 + r) * 16 + j] + Pt[(3 * 4 + r) * 16 + j]);
            if (g * 4 + r < rows) sV[(size_t)(g * 4 + r) * 16 + j] = (bf16_t)(v * salpha);
        }
    }
}

template <typename T>
static bool ln_vec_ok(int C, const void* a, const void* b, const void* c) {
    constexpr int VEC = 16 / sizeof(T);
    uintptr_t m = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c);
    return (C % VEC == 0) && (C <= 64 * VEC * 4) && ((m & 15) == 0);
}

extern "C" int cvft_layernorm_fwd(int dtype, int rows, int C, const void* x, const float* gamma, const float* beta,
                                  float eps, int relu, float post_scale, void* y, float* mean, float* rstd,
                                  void* stream) {
    CVFT_CHECK_ARG(dtype == CVFT_F32 || dtype == CVFT_BF16, "cvft_layernorm_fwd: bad dtype");
    CVFT_CHECK_ARG(rows >= 0 && C > 0 && x && gamma && beta && y && mean && rstd, "cvft_layernorm_fwd: bad args");
    if (rows == 0) return 0;
    dim3 grid((rows + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
#define LN_FWD(TT, VPv) hipLaunchKernelGGL((ln_fwd_kernel<TT, VPv>), grid, dim3(256), 0, st, rows, C, (const TT*)x, gamma, beta, eps, \
                                           relu, post_scale, (TT*)y, mean, rstd)
    if (dtype == CVFT_F32) { if (ln_vec_ok<float>(C, x, y, y)) LN_FWD(float, true); else LN_FWD(float, false); }
    else { if (ln_vec_ok<bf16_t>(C, x, y, y)) LN_FWD(bf16_t, true); else LN_FWD(bf16_t, false); }
#undef LN_FWD
    CVFT_LAUNCH_CHECK("cvft_layernorm_fwd");
    return 0;
}

extern "C" int cvft_layernorm_bwd(int dtype, int rows, int C, const void* x, const float* gamma, const float* beta,
                                  const float* mean, const float* rstd, int relu, float post_scale, const void* dy,
                                  const void* dres, void* dx, void* stream) {
    CVFT_CHECK_ARG(dtype == CVFT_F32 || dtype == CVFT_BF16, "cvft_layernorm_bwd: bad dtype");
    CVFT_CHECK_ARG(rows >= 0 && C > 0 && x && gamma && beta && mean && rstd && dy && dx, "cvft_layernorm_bwd: bad args");
    if (rows == 0) return 0;
    dim3 grid((rows + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
#define LN_BWD(TT, VPv, NCHv, RLv) hipLaunchKernelGGL((ln_bwd_kernel<TT, VPv, false, NCHv, RLv>), grid, dim3(256), 0, st, rows, C, (const TT*)x, \
                                                      gamma, beta, mean, rstd, relu, post_scale, (const TT*)dy, (const TT*)dres, (TT*)dx)
    const bool ra = (reinterpret_cast<uintptr_t>(dres) & 15) == 0;
    if (dtype == CVFT_F32) { if (ra && ln_vec_ok<float>(C, x, dy, dx)) LN_BWD(float, true, 4, true); else LN_BWD(float, false, 4, true); }
    else if (ra && ln_vec_ok<bf16_t>(C, x, dy, dx)) {
        if (relu) LN_BWD(bf16_t, true, 4, true);
        else if (C <= 1024) LN_BWD(bf16_t, true, 2, false);
        else LN_BWD(bf16_t, true, 4, false);
    } else LN_BWD(bf16_t, false, 4, true);
#undef LN_BWD
    CVFT_LAUNCH_CHECK("cvft_layernorm_bwd");
    return 0;
}

extern "C" int cvft_layernorm_bwd_mask(int dtype, int rows, int C, const void* x, const float* gamma, const float* beta,
                                       const float* mean, const float* rstd, const void* dy, const void* dres, void* dx,
                                       float p, const int64_t* seed, unsigned site, void* dxm, void* stream) {
    CVFT_CHECK_ARG(dtype == CVFT_F32 || dtype == CVFT_BF16, "cvft_layernorm_bwd_mask: bad dtype");
    CVFT_CHECK_ARG(rows >= 0 && C > 0 && x && gamma && beta && mean && rstd && dy && dx && dxm && seed, "cvft_layernorm_bwd_mask: bad args");
    CVFT_CHECK_ARG(p > 0.f && cvft_drop_rate_ok(p), "cvft_layernorm_bwd_mask: p outside [2^-16, 1 - 2^-16]");
    const bool ra = ((reinterpret_cast<uintptr_t>(dres) | reinterpret_cast<uintptr_t>(dxm)) & 15) == 0;
    const bool ok = ra && (dtype == CVFT_F32 ? ln_vec_ok<float>(C, x, dy, dx) : ln_vec_ok<bf16_t>(C, x, dy, dx));
    CVFT_CHECK_ARG(ok, "cvft_layernorm_bwd_mask: needs the vector path (C % (16 / sizeof(T)) == 0, C <= 256 chunks, 16-byte aligned pointers)");
    if (rows == 0) return 0;
    dim3 grid((rows + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == CVFT_F32)
        hipLaunchKernelGGL((ln_bwd_kernel<float, true, true>), grid, dim3(256), 0, st, rows, C, (const float*)x, gamma, beta, mean, rstd, 0, 1.f,
                           (const float*)dy, (const float*)dres, (float*)dx, p, (const long long*)seed, site, (float*)dxm);
    else if (C <= 1024)
        hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, true, true, 2, false>), grid, dim3(256), 0, st, rows, C, (const bf16_t*)x, gamma, beta, mean, rstd, 0, 1.f,
                           (const bf16_t*)dy, (const bf16_t*)dres, (bf16_t*)dx, p, (const long long*)seed, site, (bf16_t*)dxm);
    else
        hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, true, true, 4, false>), grid, dim3(256), 0, st, rows, C, (const bf16_t*)x, gamma, beta, mean, rstd, 0, 1.f,
                           (const bf16_t*)dy, (const bf16_t*)dres, (bf16_t*)dx, p, (const long long*)seed, site, (bf16_t*)dxm);
    CVFT_LAUNCH_CHECK("cvft_layernorm_bwd_mask");
    return 0;
}

extern "C" int cvft_layernorm_bwd_mask_side(int rows, int C, const void* x, const float* gamma, const float* beta,
                                            const float* mean, const float* rstd, const void* dy, const void* dres, void* dx,
                                            float p, const int64_t* seed, unsigned site, void* dxm,
                                            const void* Bt, int R, float alpha, void* V, void* stream) {
    CVFT_CHECK_ARG(rows >= 0 && C > 0 && x && gamma && beta && mean && rstd && dy && dx && dxm && seed && Bt && V,
                   "cvft_layernorm_bwd_mask_side: bad args");
    CVFT_CHECK_ARG(R == 16, "cvft_layernorm_bwd_mask_side: rank 16 only");
    CVFT_CHECK_ARG(p > 0.f && cvft_drop_rate_ok(p), "cvft_layernorm_bwd_mask_side: p outside [2^-16, 1 - 2^-16]");
    const bool ra = ((reinterpret_cast<uintptr_t>(dres) | reinterpret_cast<uintptr_t>(dxm) | reinterpret_cast<uintptr_t>(Bt)) & 15) == 0;
    CVFT_CHECK_ARG(ra && ln_vec_ok<bf16_t>(C, x, dy, dx),
                   "cvft_layernorm_bwd_mask_side: needs the vector path (bf16, C % 8 == 0, C <= 2048, 16-byte aligned pointers)");
    CVFT_CHECK_ARG(C % 128 == 0 && C <= 1536,
                   "cvft_layernorm_bwd_mask_side: C must be a multiple of 128 (a quarter of the row per wave, 32 columns per MFMA) and <= 1536 (LDS)");
    if (rows == 0) return 0;
    const int groups = (rows + 3) / 4, per = (groups + 1023) / 1024;          // equal shares: no block walks one group more than another needs to
    const size_t lds = (size_t)20 * (C + 8) * 2 + 4 * 4 * 16 * 4;             // B^T, four rows, four partial tiles
#define LN_SIDE(NCHv) hipLaunchKernelGGL(ln_bwd_side_kernel<NCHv>, dim3((groups + per - 1) / per), dim3(256), lds,                       \
                                         (hipStream_t)stream, rows, C, (const bf16_t*)x, gamma, beta, mean, rstd, (const bf16_t*)dy,    \
                                         (const bf16_t*)dres, (bf16_t*)dx, p, (const long long*)seed, site, (bf16_t*)dxm,              \
                                         (const bf16_t*)Bt, alpha, (bf16_t*)V)
    if (C <= 1024) LN_SIDE(2); else LN_SIDE(4);          // chunks of 8 columns per lane: 2 cover C <= 1024 with half the registers
#undef LN_SIDE
    CVFT_LAUNCH_CHECK("cvft_layernorm_bwd_mask_side");
    return 0;
}

// ------------------------------------------------------------------ GroupNorm + Mish
// stats: one block per (b, g); the group's (T x Cg) slab is walked in 16-byte chunks (Cg % VEC == 0) so that
// each frame's Cg-channel run is read with full-width loads.
template <typename T, bool VECP>
__global__ void __launch_bounds__(512) gn_stats_kernel(int T_, int C, int G, const T* __restrict__ x, float eps,
                                                        float* __restrict__ mean, float* __restrict__ rstd,
                                                        const int* __restrict__ t_eff) {
    constexpr int VEC = VECP ? 16 / sizeof(T) : 1;
    __shared__ float sm[16];
    const int b = blockIdx.x / G, g = blockIdx.x % G;
    const int Cg = C / G, CgV = Cg / VEC;
    const T* xb = x + (size_t)b * T_ * C + g * Cg;
    const int Te = t_eff ? min(*t_eff, T_) : T_;                   // frames the statistics run over (cvft.h: t_eff)
    const int nch = Te * CgV;
    const float n = (float)Te * (float)Cg;
    float s = 0.f;
    for (int e = threadIdx.x; e < nch; e += 512) {
        const T* p = xb + (size_t)(e / CgV) * C + (e % CgV) * VEC;
        if (VECP) {
            uint4 v = *reinterpret_cast<const uint4*>(p);
            const T* ve = reinterpret_cast<const T*>(&v);
#pragma unroll
            for (int k = 0; k < VEC; ++k) s += to_f32(ve[k]);
        } else {
            s += to_f32(*p);
        }
    }
    const float mu = block_sum(s, sm) / n;
    float v2 = 0.f;
    for (int e = threadIdx.x; e < nch; e += 512) {
        const T* p = xb + (size_t)(e / CgV) * C + (e % CgV) * VEC;
        if (VECP) {
            uint4 v = *reinterpret_cast<const uint4*>(p);
            const T* ve = reinterpret_cast<const T*>(&v);
#pragma unroll
            for (int k = 0; k < VEC; ++k) { float d = to_f32(ve[k]) - mu; v2 += d * d; }
        } else {
            float d = to_f32(*p) - mu;
            v2 += d * d;
        }
    }
    const float var = block_sum(v2, sm) / n;
    if (threadIdx.x == 0) {
        mean[blockIdx.x] = mu;
        rstd[blockIdx.x] = 1.0f / sqrtf(var + eps);
    }
}

template <typename T, bool VECP>
__global__ void __launch_bounds__(256) gn_apply_fwd_kernel(int B, int T_, int C, int G, const T* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const int* __restrict__ len,
                                                            const T* __restrict__ add, int apply_mish,
                                                            T* __restrict__ y, const int* __restrict__ t_eff) {
    // one 16-byte chunk (VEC channels of one frame, all in one group since Cg % VEC == 0) per thread and step
    constexpr int VEC = VECP ? 16 / sizeof(T) : 1;
    const size_t total = (size_t)B * T_ * C / VEC;
    const int Cg = C / G, CV = C / VEC;
    for (size_t ch = (size_t)blockIdx.x * 256 + threadIdx.x; ch < total; ch += (size_t)gridDim.x * 256) {
        const int c0 = (int)(ch % CV) * VEC;
        const size_t bt = ch / CV;
        const int t = (int)(bt % T_), b = (int)(bt / T_);
        const int sg = b * G + c0 / Cg;
        const float mu = mean[sg], rs = rstd[sg];
        const bool pad = t_eff && t >= *t_eff;                     // bucket padding: no such frame in the exact-shape batch
        const bool dead = (len && t >= len[b]) || pad;
        const size_t i = ch * VEC;
        T xv[VEC], av[VEC], ov[VEC];
        if (VECP) {
            *reinterpret_cast<uint4*>(xv) = *reinterpret_cast<const uint4*>(x + i);
            if (add) *reinterpret_cast<uint4*>(av) = *reinterpret_cast<const uint4*>(add + (size_t)b * C + c0);
        } else {
            xv[0] = x[i];
            if (add) av[0] = add[(size_t)b * C + c0];
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const int c = c0 + k;
            const float z = (to_f32(xv[k]) - mu) * rs * gamma[c] + beta[c];
            float o = apply_mish ? act_apply(CVFT_ACT_MISH, z) : z;
            if (dead) o = 0.f;
            if (add && !pad) o += to_f32(av[k]);
            ov[k] = from_f32<T>(o);
        }
        if (VECP) *reinterpret_cast<uint4*>(y + i) = *reinterpret_cast<const uint4*>(ov);
        else y[i] = ov[0];
    }
}

// backward stats: s1 = sum(gamma*dz), s2 = sum(gamma*dz*xhat) over the group.  One block per (b, g) left half the chip idle
// and walked T * Cg elements through the Mish derivative serially (24 us per launch at [16, 500, 256]): the frames are
// split into CVFT_GN_SPLIT chunks, block (b*G + g, s) writes its partial pair to ws[(sg * SPLIT + s) * 2], and the apply
// kernel adds the SPLIT partials in a fixed order (deterministic, no atomics).
template <typename T, bool VECP>
__global__ void __launch_bounds__(256) gn_bwd_stats_kernel(int T_, int C, int G, const T* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const int* __restrict__ len,
                                                            int apply_mish, const T* __restrict__ dy,
                                                            float* __restrict__ ws, const int* __restrict__ t_eff) {
    constexpr int VEC = VECP ? 16 / sizeof(T) : 1;
    __shared__ float sm[16];
    const int sg = blockIdx.x, b = sg / G, g = sg % G, sp = blockIdx.y;
    const int Cg = C / G, CgV = Cg / VEC;
    const size_t off = (size_t)b * T_ * C + g * Cg;
    const int Te = t_eff ? min(*t_eff, T_) : T_;
    const float n = (float)Te * (float)Cg;
    const float mu = mean[sg], rs = rstd[sg];
    const int lb = len ? min(len[b], Te) : Te;                     // frames t >= len contribute nothing
    const int tc = (Te + CVFT_GN_SPLIT - 1) / CVFT_GN_SPLIT;       // (chunks of the exact frame count: same partial sums as the exact-shape launch)
    const int t0 = sp * tc, t1 = min(lb, t0 + tc);
    const int nch = t1 > t0 ? (t1 - t0) * CgV : 0;
    float s1 = 0.f, s2 = 0.f;
    for (int e = threadIdx.x; e < nch; e += 256) {
        const int t = t0 + e / CgV, cc = (e % CgV) * VEC;
        const size_t i = off + (size_t)t * C + cc;
        T xv[VEC], dv[VEC];
        if (VECP) {
            *reinterpret_cast<uint4*>(xv) = *reinterpret_cast<const uint4*>(x + i);
            *reinterpret_cast<uint4*>(dv) = *reinterpret_cast<const uint4*>(dy + i);
        } else {
            xv[0] = x[i];
            dv[0] = dy[i];
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const int c = g * Cg + cc + k;
            float xh = (to_f32(xv[k]) - mu) * rs;
            float dz = to_f32(dv[k]);
            if (apply_mish) dz *= act_grad(CVFT_ACT_MISH, xh * gamma[c] + beta[c]);
            dz *= gamma[c];
            s1 += dz;
            s2 += dz * xh;
        }
    }
    s1 = block_sum(s1, sm);
    s2 = block_sum(s2, sm);
    if (threadIdx.x == 0) {
        ws[((size_t)sg * CVFT_GN_SPLIT + sp) * 2 + 0] = s1 / n;
        ws[((size_t)sg * CVFT_GN_SPLIT + sp) * 2 + 1] = s2 / n;
    }
}

template <typename T, bool VECP>
__global__ void __launch_bounds__(256) gn_apply_bwd_kernel(int B, int T_, int C, int G, const T* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const int* __restrict__ len,
                                                            int apply_mish, const T* __restrict__ dy,
                                                            const float* __restrict__ ws, T* __restrict__ dx,
                                                            const int* __restrict__ t_eff) {
    constexpr int VEC = VECP ? 16 / sizeof(T) : 1;
    const size_t total = (size_t)B * T_ * C / VEC;
    const int Cg = C / G, CV = C / VEC;
    for (size_t ch = (size_t)blockIdx.x * 256 + threadIdx.x; ch < total; ch += (size_t)gridDim.x * 256) {
        const int c0 = (int)(ch % CV) * VEC;
        const size_t bt = ch / CV;
        const int t = (int)(bt % T_), b = (int)(bt / T_);
        const int sg = b * G + c0 / Cg;
        const float mu = mean[sg], rs = rstd[sg];
        float w1 = 0.f, w2 = 0.f;
#pragma unroll
        for (int sp = 0; sp < CVFT_GN_SPLIT; ++sp) {
            const float2 pr = *reinterpret_cast<const float2*>(ws + ((size_t)sg * CVFT_GN_SPLIT + sp) * 2);
            w1 += pr.x;
            w2 += pr.y;
        }
        const bool pad = t_eff && t >= *t_eff;
        const bool live = (!len || t < len[b]) && !pad;
        const size_t i = ch * VEC;
        T xv[VEC], dv[VEC], ov[VEC];
        if (VECP) {
            *reinterpret_cast<uint4*>(xv) = *reinterpret_cast<const uint4*>(x + i);
            *reinterpret_cast<uint4*>(dv) = *reinterpret_cast<const uint4*>(dy + i);
        } else {
            xv[0] = x[i];
            dv[0] = dy[i];
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const int c = c0 + k;
            const float xh = (to_f32(xv[k]) - mu) * rs;
            float dz = 0.f;
            if (live) {
                dz = to_f32(dv[k]);
                if (apply_mish) dz *= act_grad(CVFT_ACT_MISH, xh * gamma[c] + beta[c]);
                dz *= gamma[c];
            }
            ov[k] = from_f32<T>(pad ? 0.f : rs * (dz - w1 - xh * w2));
        }
        if (VECP) *reinterpret_cast<uint4*>(dx + i) = *reinterpret_cast<const uint4*>(ov);
        else dx[i] = ov[0];
    }
}

// ------------------------------------------------------------------ GroupNorm, one launch each way
// The group's (T x Cg) slab of one utterance is small (T = 500, Cg = 32: 32 KB in bf16): ONE block per (b, g) keeps it in
// registers (up to GN_NCH 16-byte chunks per thread), forms the two-pass statistics and applies them -- x is read once
// instead of three times and the statistics launch disappears from the estimator's dependent chain (74 launches per
// Flow chain and step, forward + backward).  Chunk e = tid + 512 k walks the slab frame-major, so a bucket-padded launch
// (t_eff < T) gives every thread the same valid chunks as the exact-shape launch: bit-identical on the valid frames.
constexpr int GN_NCH = 8;

// MISH as a template flag and gamma / beta fetched per chunk BEFORE the element loop: with the run-time flag around them the
// compiler kept one 4-byte load and one branch per element (128 / 368 scalar loads and 100-300 branches per thread in the
// forward / backward kernels; a block is one of only B * G = 64 per launch, so its own instruction stream is the launch time).
template <typename T, bool MISH, int NCH>
__global__ void __launch_bounds__(512) gn_fused_fwd_kernel(int T_, int C, int G, const T* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float eps, const int* __restrict__ len, const T* __restrict__ add,
                                                            T* __restrict__ y, float* __restrict__ mean,
                                                            float* __restrict__ rstd, const int* __restrict__ t_eff) {
    constexpr int VEC = 16 / sizeof(T);
    __shared__ float sm[16];
    const int b = blockIdx.x / G, g = blockIdx.x % G;
    const int Cg = C / G, CgV = Cg / VEC;
    const size_t off = (size_t)b * T_ * C + g * Cg;
    const int Te = t_eff ? min(*t_eff, T_) : T_;
    const int nst = Te * CgV, nall = T_ * CgV;
    const float n = (float)Te * (float)Cg;
    uint4 xr[NCH];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int e = threadIdx.x + k * 512;
        xr[k] = make_uint4(0, 0, 0, 0);
        if (e < nall) xr[k] = *reinterpret_cast<const uint4*>(x + off + (size_t)(e / CgV) * C + (e % CgV) * VEC);
        if (e < nst) {
            const T* ve = reinterpret_cast<const T*>(&xr[k]);
#pragma unroll
            for (int q = 0; q < VEC; ++q) s += to_f32(ve[q]);
        }
    }
    const float mu = block_sum(s, sm) / n;
    float v2 = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int e = threadIdx.x + k * 512;
        if (e < nst) {
            const T* ve = reinterpret_cast<const T*>(&xr[k]);
#pragma unroll
            for (int q = 0; q < VEC; ++q) { const float d = to_f32(ve[q]) - mu; v2 += d * d; }
        }
    }
    const float rs = 1.0f / sqrtf(block_sum(v2, sm) / n + eps);
    if (threadIdx.x == 0) {
        mean[blockIdx.x] = mu;
        rstd[blockIdx.x] = rs;
    }
    const int lb = len ? len[b] : T_;
    // 512 % CgV == 0 (the estimator: CgV = 4): a thread's chunks all sit on the same 8 channels -- one gamma / beta fetch
    const bool same_c = (512 % CgV) == 0;
    float gm[VEC], bt[VEC];
    {
        const int c0 = g * Cg + (threadIdx.x % CgV) * VEC;
#pragma unroll
        for (int q = 0; q < VEC; ++q) { gm[q] = gamma[c0 + q]; bt[q] = beta[c0 + q]; }
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int e = threadIdx.x + k * 512;
        if (e >= nall) continue;
        const int t = e / CgV, cc = (e % CgV) * VEC, c0 = g * Cg + cc;
        const bool pad = t_eff && t >= Te;
        const bool dead = t >= lb || pad;
        const T* ve = reinterpret_cast<const T*>(&xr[k]);
        T av[VEC], ov[VEC];
        *reinterpret_cast<uint4*>(av) = add ? *reinterpret_cast<const uint4*>(add + (size_t)b * C + c0) : make_uint4(0, 0, 0, 0);
        if (!same_c) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) { gm[q] = gamma[c0 + q]; bt[q] = beta[c0 + q]; }
        }
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            const float z = (to_f32(ve[q]) - mu) * rs * gm[q] + bt[q];
            float o = MISH ? act_apply(CVFT_ACT_MISH, z) : z;
            o = dead ? 0.f : o;
            o += pad ? 0.f : to_f32(av[q]);
            ov[q] = from_f32<T>(o);
        }
        *reinterpret_cast<uint4*>(y + off + (size_t)t * C + cc) = *reinterpret_cast<const uint4*>(ov);
    }
}

template <typename T, bool MISH, int NCH>
__global__ void __launch_bounds__(512) gn_fused_bwd_kernel(int T_, int C, int G, const T* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const int* __restrict__ len,
                                                            const T* __restrict__ dy, T* __restrict__ dx,
                                                            const int* __restrict__ t_eff) {
    constexpr int VEC = 16 / sizeof(T);
    __shared__ float sm[16];
    const int b = blockIdx.x / G, g = blockIdx.x % G;
    const int Cg = C / G, CgV = Cg / VEC;
    const size_t off = (size_t)b * T_ * C + g * Cg;
    const int Te = t_eff ? min(*t_eff, T_) : T_;
    const int nall = T_ * CgV;
    const float n = (float)Te * (float)Cg;
    const float mu = mean[blockIdx.x], rs = rstd[blockIdx.x];
    const int lb = len ? min(len[b], Te) : Te;                     // frames t >= len (or in the bucket padding) contribute nothing
    const int nlive = lb * CgV;
    // dz = dy * mish'(z) * gamma is formed ONCE and kept (fp32, in place of the dy chunk) for the second pass
    uint4 xr[NCH], dr[NCH];
    float dzr[NCH][VEC];
    float s1 = 0.f, s2 = 0.f;
    const bool same_c = (512 % CgV) == 0;          // a thread's chunks all sit on the same channels: one gamma / beta fetch
    float gm[VEC], bt[VEC];
    {
        const int c0 = g * Cg + (threadIdx.x % CgV) * VEC;
#pragma unroll
        for (int q = 0; q < VEC; ++q) { gm[q] = gamma[c0 + q]; bt[q] = MISH ? beta[c0 + q] : 0.f; }
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int e = threadIdx.x + k * 512;
        xr[k] = make_uint4(0, 0, 0, 0);
        dr[k] = make_uint4(0, 0, 0, 0);
        if (e < nall) {
            const size_t i = off + (size_t)(e / CgV) * C + (e % CgV) * VEC;
            xr[k] = *reinterpret_cast<const uint4*>(x + i);
            dr[k] = *reinterpret_cast<const uint4*>(dy + i);
        }
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int e = threadIdx.x + k * 512;
        const bool live = e < nlive;
        if (!same_c) {
            const int c0 = g * Cg + (e % CgV) * VEC;
#pragma unroll
            for (int q = 0; q < VEC; ++q) { gm[q] = gamma[c0 + q]; bt[q] = MISH ? beta[c0 + q] : 0.f; }
        }
        const T* ve = reinterpret_cast<const T*>(&xr[k]);
        const T* de = reinterpret_cast<const T*>(&dr[k]);
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            const float xh = (to_f32(ve[q]) - mu) * rs;
            float dz = to_f32(de[q]);
            if (MISH) dz *= act_grad(CVFT_ACT_MISH, xh * gm[q] + bt[q]);
            dz = live ? dz * gm[q] : 0.f;
            dzr[k][q] = dz;
            s1 += dz;
            s2 += live ? dz * xh : 0.f;      // select, not 0 * xh: x of a frame the forward never normalised (bucket padding) may be anything
        }
        __builtin_amdgcn_sched_barrier(0);      // one chunk's temporaries at a time (interleaved, the 8-chunk form spilled)
    }
    const float w1 = block_sum(s1, sm) / n;
    const float w2 = block_sum(s2, sm) / n;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int e = threadIdx.x + k * 512;
        if (e >= nall) continue;
        const int t = e / CgV, cc = (e % CgV) * VEC;
        const bool pad = t_eff && t >= Te;
        const T* ve = reinterpret_cast<const T*>(&xr[k]);
        T ov[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            const float xh = (to_f32(ve[q]) - mu) * rs;
            ov[q] = from_f32<T>(pad ? 0.f : rs * (dzr[k][q] - w1 - xh * w2));
        }
        *reinterpret_cast<uint4*>(dx + off + (size_t)t * C + cc) = *reinterpret_cast<const uint4*>(ov);
    }
}

// one block per (b, g) holds the slab: vector path, at most 512 * GN_NCH chunks (CVFT_GN_FUSED=0: the two-launch kernels)
static bool gn_fused_ok(int T, int Cg, int vec, bool vec_ok) {
    static const int on = getenv("CVFT_GN_FUSED") ? atoi(getenv("CVFT_GN_FUSED")) : 1;
    return on && vec_ok && (long)T * (Cg / vec) <= 512L * GN_NCH;
}

static inline unsigned ew_grid(size_t total) {
    size_t g = (total + 255) / 256;
    return (unsigned)(g > 4096 ? 4096 : (g == 0 ? 1 : g));
}

extern "C" int cvft_groupnorm_mish_fwd(int dtype, int B, int T, int C, int G, const void* x, const float* gamma,
                                       const float* beta, float eps, const int32_t* len, const void* add,
                                       int apply_mish, void* y, float* mean, float* rstd, const int32_t* t_eff, void* stream) {
    CVFT_CHECK_ARG(dtype == CVFT_F32 || dtype == CVFT_BF16, "cvft_groupnorm_mish_fwd: bad dtype");
    CVFT_CHECK_ARG(B > 0 && T > 0 && C > 0 && G > 0 && C % G == 0, "cvft_groupnorm_mish_fwd: bad dims B%d T%d C%d G%d", B, T, C, G);
    CVFT_CHECK_ARG(x && gamma && beta && y && mean && rstd, "cvft_groupnorm_mish_fwd: null operand");
    hipStream_t st = (hipStream_t)stream;
    size_t total = (size_t)B * T * C;
    const int vec = dtype == CVFT_F32 ? 4 : 8;
    const bool vp = ((C / G) % vec == 0) && (C % vec == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    const bool vpa = vp && ((reinterpret_cast<uintptr_t>(y) & 15) == 0) && (!add || ((reinterpret_cast<uintptr_t>(add) & 15) == 0));
    if (gn_fused_ok(T, C / G, vec, vpa)) {
        // chunks per thread the kernel is unrolled for: 4 cover the estimator's slabs (500 frames x 4 chunks) with half the registers
        const bool half = (long)T * (C / G / vec) <= 512L * (GN_NCH / 2);
#define GN_FF(TT, MI) do { if (half) hipLaunchKernelGGL((gn_fused_fwd_kernel<TT, MI, GN_NCH / 2>), dim3(B * G), dim3(512), 0, st, T, C, G, (const TT*)x, gamma, beta, eps, \
                                         len, (const TT*)add, (TT*)y, mean, rstd, t_eff);                                                   \
                           else hipLaunchKernelGGL((gn_fused_fwd_kernel<TT, MI, GN_NCH>), dim3(B * G), dim3(512), 0, st, T, C, G, (const TT*)x, gamma, beta, eps, \
                                         len, (const TT*)add, (TT*)y, mean, rstd, t_eff); } while (0)
        if (dtype == CVFT_F32) { if (apply_mish) GN_FF(float, true); else GN_FF(float, false); }
        else { if (apply_mish) GN_FF(bf16_t, true); else GN_FF(bf16_t, false); }
#undef GN_FF
        CVFT_LAUNCH_CHECK("cvft_groupnorm_mish_fwd");
        return 0;
    }
    if (dtype == CVFT_F32) {
        if (vp) hipLaunchKernelGGL((gn_stats_kernel<float, true>), dim3(B * G), dim3(512), 0, st, T, C, G, (const float*)x, eps, mean, rstd, t_eff);
        else hipLaunchKernelGGL((gn_stats_kernel<float, false>), dim3(B * G), dim3(512), 0, st, T, C, G, (const float*)x, eps, mean, rstd, t_eff);
        if (vpa) hipLaunchKernelGGL((gn_apply_fwd_kernel<float, true>), dim3(ew_grid(total / vec)), dim3(256), 0, st, B, T, C, G,
                                    (const float*)x, gamma, beta, mean, rstd, len, (const float*)add, apply_mish, (float*)y, t_eff);
        else hipLaunchKernelGGL((gn_apply_fwd_kernel<float, false>), dim3(ew_grid(total)), dim3(256), 0, st, B, T, C, G,
                                (const float*)x, gamma, beta, mean, rstd, len, (const float*)add, apply_mish, (float*)y, t_eff);
    } else {
        if (vp) hipLaunchKernelGGL((gn_stats_kernel<bf16_t, true>), dim3(B * G), dim3(512), 0, st, T, C, G, (const bf16_t*)x, eps, mean, rstd, t_eff);
        else hipLaunchKernelGGL((gn_stats_kernel<bf16_t, false>), dim3(B * G), dim3(512), 0, st, T, C, G, (const bf16_t*)x, eps, mean, rstd, t_eff);
        if (vpa) hipLaunchKernelGGL((gn_apply_fwd_kernel<bf16_t, true>), dim3(ew_grid(total / vec)), dim3(256), 0, st, B, T, C, G,
                                    (const bf16_t*)x, gamma, beta, mean, rstd, len, (const bf16_t*)add, apply_mish, (bf16_t*)y, t_eff);
        else hipLaunchKernelGGL((gn_apply_fwd_kernel<bf16_t, false>), dim3(ew_grid(total)), dim3(256), 0, st, B, T, C, G,
                                (const bf16_t*)x, gamma, beta, mean, rstd, len, (const bf16_t*)add, apply_mish, (bf16_t*)y, t_eff);
    }
    CVFT_LAUNCH_CHECK("cvft_groupnorm_mish_fwd");
    return 0;
}

extern "C" int cvft_groupnorm_mish_bwd(int dtype, int B, int T, int C, int G, const void* x, const float* gamma,
                                       const float* beta, const float* mean, const float* rstd, const int32_t* len,
                                       int apply_mish, const void* dy, void* dx, float* ws, const int32_t* t_eff, void* stream) {
    CVFT_CHECK_ARG(dtype == CVFT_F32 || dtype == CVFT_BF16, "cvft_groupnorm_mish_bwd: bad dtype");
    CVFT_CHECK_ARG(B > 0 && T > 0 && C > 0 && G > 0 && C % G == 0, "cvft_groupnorm_mish_bwd: bad dims");
    CVFT_CHECK_ARG(x && gamma && beta && mean && rstd && dy && dx && ws, "cvft_groupnorm_mish_bwd: null operand");
    hipStream_t st = (hipStream_t)stream;
    size_t total = (size_t)B * T * C;
    const int vec = dtype == CVFT_F32 ? 4 : 8;
    const bool vp = ((C / G) % vec == 0) && (C % vec == 0) &&
                    (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0);
    if (gn_fused_ok(T, C / G, vec, vp)) {
        const bool half = (long)T * (C / G / vec) <= 512L * (GN_NCH / 2);
#define GN_FB(TT, MI) do { if (half) hipLaunchKernelGGL((gn_fused_bwd_kernel<TT, MI, GN_NCH / 2>), dim3(B * G), dim3(512), 0, st, T, C, G, (const TT*)x, gamma, beta, mean, \
                                         rstd, len, (const TT*)dy, (TT*)dx, t_eff);                                                         \
                           else hipLaunchKernelGGL((gn_fused_bwd_kernel<TT, MI, GN_NCH>), dim3(B * G), dim3(512), 0, st, T, C, G, (const TT*)x, gamma, beta, mean, \
                                         rstd, len, (const TT*)dy, (TT*)dx, t_eff); } while (0)
        if (dtype == CVFT_F32) { if (apply_mish) GN_FB(float, true); else GN_FB(float, false); }
        else { if (apply_mish) GN_FB(bf16_t, true); else GN_FB(bf16_t, false); }
#undef GN_FB
        CVFT_LAUNCH_CHECK("cvft_groupnorm_mish_bwd");
        return 0;
    }
    const dim3 sgrid(B * G, CVFT_GN_SPLIT);
#define GN_BWD(TT, VP)                                                                                                         \
    do {                                                                                                                       \
        hipLaunchKernelGGL((gn_bwd_stats_kernel<TT, VP>), sgrid, dim3(256), 0, st, T, C, G, (const TT*)x, gamma, beta, mean, rstd, len, \
                           apply_mish, (const TT*)dy, ws, t_eff);                                                              \
        hipLaunchKernelGGL((gn_apply_bwd_kernel<TT, VP>), dim3(ew_grid(total / (VP ? vec : 1))), dim3(256), 0, st, B, T, C, G,  \
                           (const TT*)x, gamma, beta, mean, rstd, len, apply_mish, (const TT*)dy, ws, (TT*)dx, t_eff);         \
    } while (0)
    if (dtype == CVFT_F32) { if (vp) GN_BWD(float, true); else GN_BWD(float, false); }
    else { if (vp) GN_BWD(bf16_t, true); else GN_BWD(bf16_t, false); }
#undef GN_BWD
    CVFT_LAUNCH_CHECK("cvft_groupnorm_mish_bwd");
    return 0;
}
