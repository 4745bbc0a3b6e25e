// gemm_common.h -- pieces shared by the tap-GEMM kernels (register-staged gemm.hip, LDS-DMA gemm_glds.hip).
#pragma once
#include "common.h"

#include <utility>
template <int... I, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

template <typename T>
struct GP {
    int M, N, K, Tm, Tin, Tout, in_stride, out_stride, out_off, ntaps;
    int tap_off[4];
    const int* in_len;
    const int* out_len;
    const T* A; int lda;
    const T* W; int ldw;
    const T* U; int ldu; int R;
    const T* Bl; int ldbl;
    const float* bias;
    float alpha;
    int act;
    T* preact; int ldp;
    const T* dact_src; int ldd; int dact;
    const T* residual; int ldr;
    T* C; int ldc;
    int vecA, vecW, vecU, vecB;   // 16-byte vector loads legal for that operand
    unsigned bytesA, bytesW, bytesU, bytesB;   // buffer extents for the hardware range check
    // fused side path: U = lora_scale * A_tile . La^T is computed inside this launch (La [R][K], R <= 16),
    // fed to the rank-R extension step and written to Uout [M][ldu] by the n-tile-0 blocks
    const T* La; int ldla; unsigned bytesL; float lora_scale; T* Uout; int fuse;
    int direct_epi;               // LDS-DMA kernels: register epilogue allowed (set by gemm_glds_launch)
    int xcd_nsplit;               // LDS-DMA kernels: XCDs across N (1 = linear tile ranges; 2/4/8 = rectangles, see kernel)
    float xdrop_p; const long long* xdrop_seed; unsigned xdrop_sites[4];      // masked rank extension (cvft.h); 0 = off
    float odrop_p; unsigned odrop_site;     // output dropout (cvft.h): C = residual + keep / (1 - p) * epi(.), seed = xdrop_seed; 0 = off
    int row_off;                  // rows in front of this launch's row 0 in the tensor the dropout masks index (a launch that computes a
                                  // row range of a larger product: gemm_p256_launch's row split); 0 otherwise
};

// keep / (1 - p) factors of the output-dropout mask for the 4-element group holding flat output element `idx` (idx % 4 == 0)
__device__ __forceinline__ void gemm_odrop4(const float p, const long long* seed, const unsigned site, const unsigned long long idx, float* v) {
    bool k4[4];
    cvft_keep4(cvft_drop_key(seed, site), idx >> 2, cvft_drop_thr(p), k4);
    const float inv = 1.f / (1.f - p);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = k4[e] ? v[e] * inv : 0.f;
}

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define CVFT_OOB 0x80000000u      // byte offset past every buffer (< 2 GiB each, host-checked): the load returns 0

// Epilogue tail shared by every tap-GEMM kernel: Cs holds the block's fp32 accumulators ([BM][BN + 4], already
// synchronised); rows are written as 16-byte coalesced segments with the bias/act/act'/residual/mask chain.
template <typename T, int BM, int BN, int NT>
__device__ __forceinline__ void gemm_epilogue_store(const GP<T>& p, const float* Cs, int m0, int n0, int tid) {
    constexpr int VEC = 16 / sizeof(T);
    constexpr int CLD = BN + 4;
    const bool ident = (p.Tm == p.M) && p.out_stride == 1 && p.out_off == 0;
    const bool vec_out = (p.N % VEC == 0) && (p.ldc % VEC == 0) && ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0) &&
                         (!p.preact || ((p.ldp % VEC == 0) && ((reinterpret_cast<uintptr_t>(p.preact) & 15) == 0))) &&
                         (!p.dact_src || ((p.ldd % VEC == 0) && ((reinterpret_cast<uintptr_t>(p.dact_src) & 15) == 0))) &&
                         (!p.residual || ((p.ldr % VEC == 0) && ((reinterpret_cast<uintptr_t>(p.residual) & 15) == 0)));
    constexpr int CPRO = BN / VEC;                 // output chunks per tile row
    for (int c = tid; c < BM * CPRO; c += NT) {
        const int row = c / CPRO, cc = c % CPRO;
        const int m = m0 + row;
        const int nb = n0 + cc * VEC;
        if (m >= p.M || nb >= p.N) continue;
        int b = 0, to = m;
        if (!ident) {
            b = m / p.Tm;
            to = (m - b * p.Tm) * p.out_stride + p.out_off;
            if (to >= p.Tout) continue;
        }
        const size_t orow = (size_t)b * p.Tout + to;
        const bool live = p.out_len ? (to < p.out_len[b]) : true;
        float v[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int n = nb + e;
            float x = Cs[row * CLD + cc * VEC + e] * p.alpha;
            if (p.bias && n < p.N) x += p.bias[n];
            v[e] = x;
        }
        if (vec_out) {
            T tmp[VEC];
            if (p.preact) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) tmp[e] = from_f32<T>(v[e]);
                *reinterpret_cast<uint4*>(&p.preact[orow * p.ldp + nb]) = *reinterpret_cast<uint4*>(tmp);
            }
            act_apply_vec<VEC>(p.act, v);
            if (p.dact_src) {
                uint4 dv = *reinterpret_cast<const uint4*>(&p.dact_src[orow * p.ldd + nb]);
                const T* de = reinterpret_cast<const T*>(&dv);
                float ds[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) ds[e] = to_f32(de[e]);
                act_grad_mul_vec<VEC>(p.dact, v, ds);
            }
            if (p.odrop_p > 0.f) {
#pragma unroll
                for (int e = 0; e < VEC; e += 4) gemm_odrop4(p.odrop_p, p.xdrop_seed, p.odrop_site, (orow + p.row_off) * (unsigned long long)p.N + nb + e, v + e);
            }
            if (p.residual) {
                uint4 rv = *reinterpret_cast<const uint4*>(&p.residual[orow * p.ldr + nb]);
                const T* re = reinterpret_cast<const T*>(&rv);
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] += to_f32(re[e]);
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) tmp[e] = from_f32<T>(live ? v[e] : 0.f);
            *reinterpret_cast<uint4*>(&p.C[orow * p.ldc + nb]) = *reinterpret_cast<uint4*>(tmp);
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int n = nb + e;
                if (n >= p.N) continue;
                float x = v[e];
                if (p.preact) p.preact[orow * p.ldp + n] = from_f32<T>(x);
                x = act_apply(p.act, x);
                if (p.dact_src) x *= act_grad(p.dact, to_f32(p.dact_src[orow * p.ldd + n]));
                if (p.odrop_p > 0.f) {
                    const unsigned long long idx = (orow + p.row_off) * (unsigned long long)p.N + n;
                    bool k4[4];
                    cvft_keep4(cvft_drop_key(p.xdrop_seed, p.odrop_site), idx >> 2, cvft_drop_thr(p.odrop_p), k4);
                    x = k4[idx & 3] ? x / (1.f - p.odrop_p) : 0.f;
                }
                if (p.residual) x += to_f32(p.residual[orow * p.ldr + n]);
                if (!live) x = 0.f;
                p.C[orow * p.ldc + n] = from_f32<T>(x);
            }
        }
    }
}

// Register epilogue for kernels whose accumulators hold 4 CONSECUTIVE output columns per lane (MFMA issued with the
// operands swapped: lane (kg, l15) owns C[m = l15][n = 4*kg .. 4*kg+3] of a 16x16 tile): the whole
// bias/act/act'/residual/mask chain runs on that 4-vector and every global access is one 8-byte (bf16) access -- no
// fp32 staging tile in LDS, no barriers.  Caller guarantees m < M, n + 3 < N and 8-byte alignment of all row pitches.
__device__ __forceinline__ void gemm_epilogue_direct4(const GP<bf16_t>& p, const f32x4& a, int m, int n) {
    int b = 0, to = m;
    if (!((p.Tm == p.M) && p.out_stride == 1 && p.out_off == 0)) {
        b = m / p.Tm;
        to = (m - b * p.Tm) * p.out_stride + p.out_off;
        if (to >= p.Tout) return;
    }
    const size_t orow = (size_t)b * p.Tout + to;
    const bool live = p.out_len ? (to < p.out_len[b]) : true;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = a[e] * p.alpha;
    if (p.bias) {
        const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
    }
    if (p.preact) {
        bf16x4 t = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        *reinterpret_cast<bf16x4*>(&p.preact[orow * p.ldp + n]) = t;
    }
    act_apply_vec<4>(p.act, v);
    if (p.dact_src) {
        const bf16x4 d = *reinterpret_cast<const bf16x4*>(&p.dact_src[orow * p.ldd + n]);
        float ds[4] = {(float)d[0], (float)d[1], (float)d[2], (float)d[3]};
        act_grad_mul_vec<4>(p.dact, v, ds);
    }
    if (p.odrop_p > 0.f) gemm_odrop4(p.odrop_p, p.xdrop_seed, p.odrop_site, (orow + p.row_off) * (unsigned long long)p.N + n, v);
    if (p.residual) {
        const bf16x4 r = *reinterpret_cast<const bf16x4*>(&p.residual[orow * p.ldr + n]);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
    }
    bf16x4 o = {(bf16_t)(live ? v[0] : 0.f), (bf16_t)(live ? v[1] : 0.f), (bf16_t)(live ? v[2] : 0.f), (bf16_t)(live ? v[3] : 0.f)};
    // (non-temporal stores here: +3 % on 5328x4096x1024, -3 % on 5328x1024x4096, -10 % on the estimator shapes, step 31.7 vs 30.7 ms)
    *reinterpret_cast<bf16x4*>(&p.C[orow * p.ldc + n]) = o;
}

// 8-wide form of the register epilogue: the LDS-DMA kernels lay the W image out so that a lane's accumulators of the
// tile pair (j, j + 1) are 8 CONSECUTIVE output columns (gemm_glds.hip, "column map"): one 16-byte access per lane and
// operand, 64 contiguous bytes per output row and wave instruction instead of 32 (measured with the stores elided: the
// 8-byte epilogue was 7.4 of 28.9 us at 2048x4096x1024 and 11.5 of 23.3 us at 8000x1536x256).
// Caller guarantees m < M, n + 7 < N, n % 8 == 0 and 16-byte alignment of every row pitch (glds_wide_epilogue).
__device__ __forceinline__ void gemm_epilogue_direct8(const GP<bf16_t>& p, const f32x4& a0, const f32x4& a1, int m, int n) {
    int b = 0, to = m;
    if (!((p.Tm == p.M) && p.out_stride == 1 && p.out_off == 0)) {
        b = m / p.Tm;
        to = (m - b * p.Tm) * p.out_stride + p.out_off;
        if (to >= p.Tout) return;
    }
    const size_t orow = (size_t)b * p.Tout + to;
    const bool live = p.out_len ? (to < p.out_len[b]) : true;
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a0[e] * p.alpha; v[4 + e] = a1[e] * p.alpha; }
    if (p.bias) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
    }
    if (p.preact) {
        bf16x8 t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (bf16_t)v[e];
        *reinterpret_cast<bf16x8*>(&p.preact[orow * p.ldp + n]) = t;
    }
    act_apply_vec<8>(p.act, v);
    if (p.dact_src) {
        const bf16x8 d = *reinterpret_cast<const bf16x8*>(&p.dact_src[orow * p.ldd + n]);
        float ds[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) ds[e] = (float)d[e];
        act_grad_mul_vec<8>(p.dact, v, ds);
    }
    if (p.odrop_p > 0.f) {
        gemm_odrop4(p.odrop_p, p.xdrop_seed, p.odrop_site, (orow + p.row_off) * (unsigned long long)p.N + n, v);
        gemm_odrop4(p.odrop_p, p.xdrop_seed, p.odrop_site, (orow + p.row_off) * (unsigned long long)p.N + n + 4, v + 4);
    }
    if (p.residual) {
        const bf16x8 r = *reinterpret_cast<const bf16x8*>(&p.residual[orow * p.ldr + n]);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)r[e];
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(live ? v[e] : 0.f);
    *reinterpret_cast<bf16x8*>(&p.C[orow * p.ldc + n]) = o;
}

// ---------------------------------------------------------------- shared by the LDS-DMA kernels (gemm_glds.hip, gemm_p256.hip)
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// register epilogue (gemm_epilogue_direct4) is legal: 8-byte aligned bf16 rows everywhere, 16-byte aligned bias
__host__ __device__ inline bool glds_direct_epilogue(const GP<bf16_t>& p) {
    return p.direct_epi && (p.N % 4 == 0) && (p.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.C) & 7) == 0) &&
           (!p.bias || ((reinterpret_cast<uintptr_t>(p.bias) & 15) == 0)) &&
           (!p.preact || ((p.ldp % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.preact) & 7) == 0))) &&
           (!p.dact_src || ((p.ldd % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.dact_src) & 7) == 0))) &&
           (!p.residual || ((p.ldr % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.residual) & 7) == 0)));
}

// register epilogue in its 8-wide form (gemm_epilogue_direct8): 16-byte aligned rows and bias
__host__ __device__ inline bool glds_wide_epilogue(const GP<bf16_t>& p) {
    return (p.N % 8 == 0) && (p.ldc % 8 == 0) && ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0) &&
           (!p.bias || ((reinterpret_cast<uintptr_t>(p.bias) & 15) == 0)) &&
           (!p.preact || ((p.ldp % 8 == 0) && ((reinterpret_cast<uintptr_t>(p.preact) & 15) == 0))) &&
           (!p.dact_src || ((p.ldd % 8 == 0) && ((reinterpret_cast<uintptr_t>(p.dact_src) & 15) == 0))) &&
           (!p.residual || ((p.ldr % 8 == 0) && ((reinterpret_cast<uintptr_t>(p.residual) & 15) == 0)));
}

// Column map of the W image.  The MFMA is issued with swapped operands, so lane (kg, l15) of output tile j owns row
// l15 and the four n-slots 4*kg .. 4*kg+3 of that tile.  Slot s of tile j is NOT column 16*j + s: it is
//     col(j, s) = 32*(j >> 1) + 8*(s >> 2) + 4*(j & 1) + (s & 3)
// so that the lane's accumulators of the tile pair (j, j+1) are the 8 consecutive columns 32*(j>>1) + 8*kg .. +7: the
// epilogue moves 16 bytes per lane and the 4 kg-lanes of a row cover 64 contiguous bytes.  Only the W side knows: the
// fragment of tile j reads image rows col(j, 0..15) (four runs of 4 rows, 8 apart), and the source-side XOR of the W
// image is wswz(r) = bit1(r) | bits3..4(r) << 1, which keeps those reads bank-conflict free (the A image keeps
// (r >> 1) & 7 for its 16 consecutive rows).
__host__ __device__ constexpr int glds_col(int j, int s) { return 32 * (j >> 1) + 8 * (s >> 2) + 4 * (j & 1) + (s & 3); }
__host__ __device__ constexpr int glds_wswz(int r) { return ((r >> 1) & 1) | (((r >> 3) & 3) << 1); }

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }


// LDS-DMA (global_load_lds) bf16 kernels for identity-geometry GEMMs; returns 1 when the shape is not eligible.
int gemm_glds_launch(const GP<bf16_t>& p, hipStream_t st, int cfg);
// rank-side products C[M, R<=64] = alpha * A W^T without epilogue (skinny.hip); returns 1 when not eligible.
int skinny_launch(const GP<bf16_t>& p, hipStream_t st);
// 256x256 tile, 8 waves, 8-phase LDS-DMA pipeline for the LLM-sized launches (gemm_p256.hip); returns 1 when not eligible.
int gemm_p256_launch(const GP<bf16_t>& p, hipStream_t st);
