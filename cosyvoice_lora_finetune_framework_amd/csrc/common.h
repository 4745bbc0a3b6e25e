// common.h -- shared device helpers for libcvft (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>
#include "../../include/cvft.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define WAVE 64

extern "C" void cvft_set_error(const char* fmt, ...);
extern "C" void cvft_set_kernel_label(const char* fmt, ...);
extern "C" int cvft_concurrent_chains(void);

#define CVFT_CHECK_ARG(cond, ...)                      \
    do {                                               \
        if (!(cond)) {                                 \
            cvft_set_error(__VA_ARGS__);               \
            return -1;                                 \
        }                                              \
    } while (0)

#define CVFT_HIP_CHECK_RET(call, name)                                          \
    do {                                                                        \
        hipError_t e__ = (call);                                                \
        if (e__ != hipSuccess) {                                                \
            cvft_set_error("%s: %s", name, hipGetErrorString(e__));             \
            return -2;                                                          \
        }                                                                       \
    } while (0)

#define CVFT_LAUNCH_CHECK(name)                                                 \
    do {                                                                        \
        hipError_t e__ = hipGetLastError();                                     \
        if (e__ != hipSuccess) {                                                \
            cvft_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return -2;                                                          \
        }                                                                       \
    } while (0)

// ---------------------------------------------------------------- conversions
__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }

// ---------------------------------------------------------------- MFMA traits
// One 16x16 output tile per wave-instruction; operands are "k-contiguous rows":
// A[row][k], B[col][k] (i.e. both row-major with k fastest), see
// cdna_hip_programming.md section 3 (A/B operand lane maps, C/D map).
//   C/D: acc[i] = C[row = (lane>>4)*4 + i][col = lane&15]
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static constexpr int K = 32;
    typedef bf16x8 Frag;
    // p points at element [row = lane&15][k0]; lane reads k0 + 8*(lane>>4) .. +7 (16 B)
    static __device__ __forceinline__ Frag load(const bf16_t* p, int lane) {
        return *reinterpret_cast<const bf16x8*>(p + 8 * (lane >> 4));
    }
    // Bank-conflict-free variant for 144-byte row pitch tiles: a ds_read_b128 is served in 16-lane groups that mix
    // chunk (lane>>4) = 0 of rows {0-3,12-15} with chunk 1 of rows {4-11} (MI355X_MICROARCH.md, LDS table), which
    // collide 2-way on a linear image.  The image is stored with the low chunk bit XOR-ed by sw(row&15), so every
    // group reads one physical chunk column of 16 distinct rows.  Writers must use chunk_sw() too.
    static __device__ __forceinline__ int sw(int row16) { return ((row16 + 4) >> 3) & 1; }
    static __device__ __forceinline__ int chunk_sw(int row, int chunk) { return chunk ^ sw(row & 15); }
    static __device__ __forceinline__ Frag load_sw(const bf16_t* p, int lane) {
        return *reinterpret_cast<const bf16x8*>(p + 8 * ((lane >> 4) ^ sw(lane & 15)));
    }
    static __device__ __forceinline__ void mma(f32x4& acc, Frag a, Frag b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    }
    static __device__ __forceinline__ Frag zero() { Frag z = {0, 0, 0, 0, 0, 0, 0, 0}; return z; }
};
template <> struct Mma<float> {
    static constexpr int K = 4;
    typedef float Frag;
    static __device__ __forceinline__ Frag load(const float* p, int lane) { return p[lane >> 4]; }
    static __device__ __forceinline__ Frag load_sw(const float* p, int lane) { return p[lane >> 4]; }
    static __device__ __forceinline__ int chunk_sw(int row, int chunk) { return chunk; }
    static __device__ __forceinline__ void mma(f32x4& acc, Frag a, Frag b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    static __device__ __forceinline__ Frag zero() { return 0.f; }
};

// ---------------------------------------------------------------- activations
// erf with |abs error| <= 1.5e-7 (Abramowitz & Stegun 7.1.26: 1 - (a1 t + .. + a5 t^5) exp(-x^2), t = 1/(1 + p|x|)):
// one v_exp_f32 + one v_rcp_f32 + 6 FMAs instead of libm erff (~35 instructions); the exact-erf GELU of the
// estimator's feed-forward runs on 4 M elements per launch in the GEMM epilogue.
__device__ __forceinline__ float cvft_erf(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(1.f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float r = 1.f - poly * __expf(-ax * ax);
    return copysignf(r, x);
}

__device__ __forceinline__ float act_apply(int act, float x) {
    switch (act) {
        case CVFT_ACT_RELU: return x > 0.f ? x : 0.f;
        case CVFT_ACT_SILU: return x / (1.f + __expf(-x));
        case CVFT_ACT_GELU_ERF: return 0.5f * x * (1.f + cvft_erf(x * 0.70710678118654752440f));
        case CVFT_ACT_GELU_TANH: {
            const float k0 = 0.79788456080286535588f, k1 = 0.044715f;
            return 0.5f * x * (1.f + tanhf(k0 * (x + k1 * x * x * x)));
        }
        case CVFT_ACT_MISH: {
            // x * tanh(softplus(x)) with tanh(log(1+e)) = ((1+e)^2 - 1) / ((1+e)^2 + 1) = n / (n + 2), n = e*(e + 2):
            // one v_exp_f32 instead of expf + log1pf + tanhf (the GroupNorm kernels were VALU-bound on those)
            // (x > 20: n / (n + 2) is exactly 1 in fp32 -- the clamp only keeps e finite, no branch per element)
            const float e = __expf(fminf(x, 20.f)), n = e * (e + 2.f);
            return x * (n / (n + 2.f));
        }
        default: return x;
    }
}
__device__ __forceinline__ float act_grad(int act, float x) {
    switch (act) {
        case CVFT_ACT_RELU: return x > 0.f ? 1.f : 0.f;
        case CVFT_ACT_SILU: {
            float s = 1.f / (1.f + __expf(-x));
            return s * (1.f + x * (1.f - s));
        }
        case CVFT_ACT_GELU_ERF: {
            float cdf = 0.5f * (1.f + cvft_erf(x * 0.70710678118654752440f));
            float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
            return cdf + x * pdf;
        }
        case CVFT_ACT_GELU_TANH: {
            const float k0 = 0.79788456080286535588f, k1 = 0.044715f;
            float x2 = x * x;
            float th = tanhf(k0 * (x + k1 * x * x2));
            return 0.5f * (1.f + th) + 0.5f * x * (1.f - th * th) * k0 * (1.f + 3.f * k1 * x2);
        }
        case CVFT_ACT_MISH: {
            const float e = __expf(fminf(x, 20.f)), n = e * (e + 2.f);      // x > 20: th == 1 exactly, the second term vanishes
            const float th = n / (n + 2.f), sg = e / (1.f + e);
            return th + x * (1.f - th * th) * sg;
        }
        default: return 1.f;
    }
}

// vector forms with the activation switch OUTSIDE the element loop (one branch per vector, not per element: the
// per-element form unrolled 8x inside the GEMM epilogues was most of those kernels' code size)
template <int N> __device__ __forceinline__ void act_apply_vec(int act, float (&v)[N]) {
    switch (act) {
#define CVFT_ACT_CASE(A) case A: _Pragma("unroll") for (int e = 0; e < N; ++e) v[e] = act_apply(A, v[e]); break;
        CVFT_ACT_CASE(CVFT_ACT_RELU) CVFT_ACT_CASE(CVFT_ACT_SILU) CVFT_ACT_CASE(CVFT_ACT_GELU_ERF)
        CVFT_ACT_CASE(CVFT_ACT_GELU_TANH) CVFT_ACT_CASE(CVFT_ACT_MISH)
#undef CVFT_ACT_CASE
        default: break;
    }
}
template <int N> __device__ __forceinline__ void act_grad_mul_vec(int act, float (&v)[N], const float (&src)[N]) {
    switch (act) {
#define CVFT_ACT_CASE(A) case A: _Pragma("unroll") for (int e = 0; e < N; ++e) v[e] *= act_grad(A, src[e]); break;
        CVFT_ACT_CASE(CVFT_ACT_RELU) CVFT_ACT_CASE(CVFT_ACT_SILU) CVFT_ACT_CASE(CVFT_ACT_GELU_ERF)
        CVFT_ACT_CASE(CVFT_ACT_GELU_TANH) CVFT_ACT_CASE(CVFT_ACT_MISH)
#undef CVFT_ACT_CASE
        default: break;
    }
}

// ---------------------------------------------------------------- counter-based dropout masks
// keep(i) for element i of a tensor is a pure function of (*seed, site, i): SplitMix64 finaliser on a 64-bit counter,
// 4 x 16 random bits per group of 4 consecutive elements.  Every kernel that needs the mask of a (seed, site) pair
// (cvft_dropout_add, the dropout-skinny product, the LoRA side dgrad) calls these, so they all see the same mask.
__device__ __forceinline__ unsigned long long cvft_mix64(unsigned long long z) {
    z += 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
__device__ __forceinline__ unsigned long long cvft_drop_key(const long long* seed, unsigned site) {
    return cvft_mix64((unsigned long long)seed[0] ^ ((unsigned long long)site << 32));
}
// 16 random bits per element: ONE draw of 64 bits serves a group of 4 consecutive elements (its four 16-bit fields).  The draw is
// two 32-bit murmur3 finalisers of the group index under the two halves of the site key: four 32-bit multiplies (quarter-rate
// integer work) against the 64-bit SplitMix finaliser's two 64 x 64 multiplies (four 32-bit multiplies EACH plus carries) --
// measured in the q|k|v chain kernels, where three masks per element are re-derived each way: the mask phases were 6.9 k (forward)
// and 5.4 k (backward) ticks of a ~50 k-tick launch.  These masks are re-derived inside GEMM epilogues,
// rank-side products and the dropout passes of every train-mode step.  p is quantised to 1/65536 (0.05 -> 0.050003), the keep
// scale stays 1 / (1 - p).  The group index is taken modulo 2^32 (tensors of up to 2^34 elements have distinct groups).
__device__ __forceinline__ unsigned cvft_fmix32(unsigned h) {
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    return h ^ (h >> 16);
}
// Rates the 16-bit fields represent without bias: p == 0 (off) or 2^-16 <= p <= 1 - 2^-16.  Below 2^-16 the threshold rounds to
// 0 (nothing dropped, everything still scaled by 1 / (1 - p)); above 1 - 2^-16 it clamps (1 in 65536 kept, scaled by a huge
// factor): every C entry that takes a dropout rate rejects those (CVFT_CHECK_ARG), see include/cvft.h "Dropout masks".
static inline bool cvft_drop_rate_ok(float p) { return p == 0.f || (p >= 1.f / 65536.f && p <= 1.f - 1.f / 65536.f); }
__device__ __forceinline__ unsigned cvft_drop_thr(float p) { return (unsigned)fminf(65535.f, rintf(p * 65536.f)); }
// keep flags of elements 4g .. 4g+3: field e of the draw >= thr
// The group index first goes through a per-site BIJECTION that is not an XOR with a constant: multiplication by an odd multiplier
// taken from the site key (one more 32-bit multiply per draw; the multiplier is loop-invariant).  With gl ^ key alone every site's
// and step's mask was one fixed 2^32-entry table read at XOR-relabelled positions: drop counts over aligned power-of-two blocks
// of elements depended on the key's high bits only, and two sites' masks were permutations of each other block for block
// (tests/test_host_logic_cpu.py::test_dropout_masks_of_two_sites_are_uncorrelated pins the host replica of this function).
__device__ __forceinline__ void cvft_keep4(unsigned long long key, unsigned long long g, unsigned thr, bool (&k)[4]) {
    const unsigned gl = (unsigned)g * ((unsigned)(key >> 17) | 1u);
    const unsigned lo = cvft_fmix32(gl ^ (unsigned)key), hi = cvft_fmix32(gl ^ (unsigned)(key >> 32));
    k[0] = (lo & 0xffffu) >= thr; k[1] = (lo >> 16) >= thr; k[2] = (hi & 0xffffu) >= thr; k[3] = (hi >> 16) >= thr;
}

// ---------------------------------------------------------------- reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// block-wide sum, blockDim.x multiple of 64, <= 1024; `sm` >= 16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* sm) {
    v = wave_sum(v);
    int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = blockDim.x >> 6;
    __syncthreads();
    if (l == 0) sm[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += sm[i];
    return r;
}
