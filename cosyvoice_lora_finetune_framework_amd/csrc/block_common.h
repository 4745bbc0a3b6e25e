// block_common.h -- shared device helpers of the row-tile chain kernels (block_fused.hip: block tail; block_qkv.hip: LayerNorm +
// q|k|v projection with LoRA, and its backward).  See block_fused.hip for the structure (weight streams, fragment ring,
// accumulator-as-operand chaining, LDS carve).
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

// Diagnostic build only (-DBF_STAMPS, tools/block_stamps.py): s_memtime at the phase boundaries of wave 0 of one block of the
// forward kernel, read back by cvft_debug_block_stamps of that build (never in the product library).
#ifdef BF_STAMPS
__device__ unsigned long long bf_stamps[32];
#define BF_STAMP_BLOCK 7
#define BF_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__) :: "memory"); \
        __builtin_amdgcn_sched_barrier(0); if (blockIdx.x == BF_STAMP_BLOCK && threadIdx.x == 0) bf_stamps[i] = t__; } while (0)
extern "C" int cvft_debug_block_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(bf_stamps), sizeof(bf_stamps)) == hipSuccess ? 0 : -1;
}
#else
#define BF_STAMP(i)
#endif

// A value loaded BEFORE a run-time loop that issues loads and first used AFTER it makes the compiler wait for every outstanding load
// at that use (it cannot count the loop's loads: s_waitcnt vmcnt(0)) -- including the ring's re-requests of the moment.  Pinning
// the value (an empty asm that "modifies" it) behind the next barrier moves the wait to where the value has long arrived.
#define BF_PIN(x) asm volatile("" : "+v"(x))
#define BF_PIN64(x) asm volatile("" : "+v"(reinterpret_cast<unsigned long long&>(x)))     /* an 8-byte vector */

#define BF_ROWS 32
#define BF_D 256
#define BF_CT (BF_D / 32)                 // 8 output-feature tiles of the residual stream
#define BF_KS (BF_D / 16)                 // 16 k-steps over the residual stream
#define BF_RING 32                        // weight fragments in flight per wave
#ifndef BF_TOUCH
#define BF_TOUCH 1                        // cold-weight prefetch (bf_touch_stream); -DBF_TOUCH=0 builds the A/B library
#endif

// LDS carve (dynamic, 16-byte aligned base)
#define BF_LDS_PART 0                     // 4 waves x [8 ct][4 g][64 lanes] f32x4 = 128 KB
#define BF_LDS_TILE (4 * 32768)           // [32 rows][256] bf16, 16-byte chunks XOR-swizzled by (row & 15): 16 KB
#define BF_LDS_STAT (BF_LDS_TILE + 16384) // [3][4 waves][32 rows] floats
#define BF_LDS_BIAS (BF_LDS_STAT + 3 * 4 * 32 * 4)   // F floats: the hidden bias (read per tile without touching vmcnt)
#define BF_MAX_F 2048
#define BF_LDS_PAR (BF_LDS_BIAS + 4 * BF_MAX_F)      // 4 x 256 floats: bo | gamma | beta | b2 (a late global load would queue
                                                     // behind the 32 KB of weight fragments in flight: vmcnt retires in order)
#define BF_LDS_TOTAL (BF_LDS_PAR + 4 * 4 * BF_D)
#define BF_LDS_DUMMY BF_LDS_TOTAL          // 256 bytes: landing area of bf_touch_stream_lds (kernels launched with BF_LDS_TOTAL_DMA)
#define BF_LDS_TOTAL_DMA (BF_LDS_TOTAL + 256)

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// GELU and its derivative (the feed-forward's activation runs on 32 x F values per row tile in a kernel with a single wave per
// SIMD: its VALU cost sits beside the MFMAs of the same wave).  erf form (diffusers GELU, approximate="none"):
// Abramowitz-Stegun 7.1.26 on u = |x| / sqrt(2), t = 1 / (1 + p u), 1 - erf(u) = poly(t) exp(-u^2), |error| <= 1.5e-7;
// Phi(x) = 1 - q (x >= 0) or q (x < 0) with q = poly(t) exp(-x^2 / 2) / 2;  gelu = x Phi,  gelu' = Phi + x phi(x).
// tanh form ("gelu-approximate"): Phi ~ sigmoid(2 k0 (x + k1 x^3)).
template <int ACT, bool GRAD>
__device__ __forceinline__ float bf_gelu(float x) {
    if (ACT == CVFT_ACT_GELU_ERF) {
        const float ax = fabsf(x);
        const float t = __builtin_amdgcn_rcpf(fmaf(0.23164190f, ax, 1.f));           // p / sqrt(2)
        float q = fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
        q = fmaf(q, t, 0.5f * 1.421413741f);
        q = fmaf(q, t, 0.5f * -0.284496736f);
        q = fmaf(q, t, 0.5f * 0.254829592f);
        const float e2 = __builtin_amdgcn_exp2f(-0.72134752f * x * x);               // exp(-x^2 / 2)
        q = q * t * e2;
        const float cdf = x >= 0.f ? 1.f - q : q;
        return GRAD ? fmaf(x * 0.39894228f, e2, cdf) : x * cdf;
    } else {
        const float k0 = 0.79788456080286535588f, k1 = 0.044715f;
        const float x2 = x * x;
        const float u = k0 * x * fmaf(k1, x2, 1.f);
        const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.88539008f * u));   // sigmoid(2u) = (1 + tanh u) / 2
        // d/dx [x sigmoid(2u)] = sg + x * 2 sg (1 - sg) * du/dx,  du/dx = k0 (1 + 3 k1 x^2)
        return GRAD ? fmaf(x * 2.f * sg * (1.f - sg), k0 * fmaf(3.f * k1, x2, 1.f), sg) : x * sg;
    }
}

// Value AND derivative in one evaluation: the forward kernels save gelu'(z) (bf16) in the workspace the backward reads, so that the
// backward's activation is one multiplication (with one wave per SIMD every VALU instruction of the activation is added to the
// wave's MFMA issue time: DESIGN.md section 12).  erf form: Phi(x) = 1/2 + xc P(xc^2), xc = clamp(x, -4, 4), P = degree-8 minimax fit
// (|error| <= 7e-6 in fp32 Horner; beyond the clamp Phi is 2.6e-5 from 0 / 1), no transcendental;  gelu = x Phi,  gelu' = Phi + x phi(x)
// with one exp2.  tanh form: as bf_gelu.
template <int ACT>
__device__ __forceinline__ void bf_gelu2(float x, float& g, float& dg) {
    if (ACT == CVFT_ACT_GELU_ERF) {
        const float xc = __builtin_amdgcn_fmed3f(x, -4.f, 4.f);
        const float u = xc * xc;
        float p = fmaf(u, 8.063223411e-11f, -7.003337503e-09f);
        p = fmaf(p, u, 2.716120992e-07f);
        p = fmaf(p, u, -6.294948650e-06f);
        p = fmaf(p, u, 9.890766180e-05f);
        p = fmaf(p, u, -1.133920192e-03f);
        p = fmaf(p, u, 9.877472248e-03f);
        p = fmaf(p, u, -6.641059018e-02f);
        p = fmaf(p, u, 3.989227081e-01f);
        const float cdf = fmaf(xc, p, 0.5f);
        g = x * cdf;
        const float e2 = __builtin_amdgcn_exp2f(-0.72134752f * x * x);               // exp(-x^2 / 2)
        dg = fmaf(x * 0.39894228f, e2, cdf);
    } else {
        const float k0 = 0.79788456080286535588f, k1 = 0.044715f;
        const float x2 = x * x;
        const float w = k0 * x * fmaf(k1, x2, 1.f);
        const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.88539008f * w));
        g = x * sg;
        dg = fmaf(x * 2.f * sg * (1.f - sg), k0 * fmaf(3.f * k1, x2, 1.f), sg);
    }
}

// Cold-weight prefetch.  In the training step a block's weight stream was last read a whole step ago: it comes from HBM, and the
// 32 KB per wave the ring keeps in flight cover an L2 hit (~0.3 us), not an HBM miss (~2 us): the stream ran at half its rate
// (35 us per launch in the step against 22 us on L2-warm operands).  So every workgroup first TOUCHES one 128-byte line in eight
// of the whole stream -- the workgroups that share an XCD (equal blockIdx % 8: speed only, never correctness) split it by
// (blockIdx / 8) % 8 -- which puts all of it in flight towards that XCD's L2 at once; the ring's loads then hit lines that are
// resident or already on their way.  The touched words are OR-ed into a value nobody reads before the kernel's end.
// The loads are requested BEHIND the ring fill and nothing reads their values before the kernel's end (bf_touch_fold): a load whose
// value is used at once makes the wave wait for every load requested before it (vmcnt retires in order), and a run-time loop of
// loads makes the compiler wait for all outstanding loads at the next use of any of them.
struct BfTouch { unsigned v[8]; };
__device__ __forceinline__ BfTouch bf_touch_stream(const void* stream, int total_frags) {
    const int part = (blockIdx.x >> 3) & 7;
    const int lines = total_frags;                     // 8 lines of 128 B per fragment, one eighth of them per workgroup
    const char* base = reinterpret_cast<const char*>(stream) + (size_t)part * lines * 128;
    BfTouch r;
#pragma unroll
    for (int i = 0; i < 8; ++i)                        // (total_frags <= 2048; surplus lanes re-touch the last line)
        r.v[i] = *reinterpret_cast<const unsigned*>(base + (size_t)min(i * 256 + (int)threadIdx.x, lines - 1) * 128);
    return r;
}
// The same touches through the LDS-DMA path: the loaded words go to a 256-byte dummy area of LDS nobody reads (every wave and every
// request lands on the same bytes), so no register waits for them -- for the kernel that has none to spare (block_link_fwd_kernel:
// with the eight values held in registers to its end the compiler spills, and a spilled load result is a wait for every load in
// flight).  lds_dummy: 256 bytes of dynamic LDS behind BF_LDS_TOTAL.
__device__ __forceinline__ void bf_touch_stream_lds(const void* stream, int total_frags, char* lds_dummy) {
    typedef __attribute__((address_space(3))) void lds_v;
    typedef const __attribute__((address_space(1))) void glb_v;
    const int part = (blockIdx.x >> 3) & 7;
    const int lines = total_frags;
    const char* base = reinterpret_cast<const char*>(stream) + (size_t)part * lines * 128;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds((glb_v*)(base + (size_t)min(i * 256 + (int)threadIdx.x, lines - 1) * 128), (lds_v*)lds_dummy, 4, 0, 0);
}
__device__ __forceinline__ unsigned bf_touch_fold(const BfTouch& r) {
    unsigned a = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) a |= r.v[i];
    return a;
}

// Sum the four waves' partial tiles acc[8] (feature tile ct, rows on the lanes) through LDS; afterwards wave w holds feature
// tiles 2w, 2w+1 of all 32 rows: v[c2][4g + i] = element (row lane&31, feature 64w + 32 c2 + 8g + 4 (lane>>5) + i).
__device__ __forceinline__ void bf_reduce(char* smem, int wave, int lane, const f32x16 (&acc)[BF_CT], float (&v)[2][16]) {
    f32x4* part = reinterpret_cast<f32x4*>(smem + BF_LDS_PART);
#pragma unroll
    for (int ct = 0; ct < BF_CT; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 t = {acc[ct][4 * g], acc[ct][4 * g + 1], acc[ct][4 * g + 2], acc[ct][4 * g + 3]};
            part[((wave * BF_CT + ct) * 4 + g) * 64 + lane] = t;
        }
    __syncthreads();
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 s = part[((0 * BF_CT + 2 * wave + c2) * 4 + g) * 64 + lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) s += part[((w * BF_CT + 2 * wave + c2) * 4 + g) * 64 + lane];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[c2][4 * g + i] = s[i];
        }
}

// per-row sum over the 256 features of a quantity each wave holds for its 64 features: lanes of both halves return the total
__device__ __forceinline__ float bf_rowsum(char* smem, int slot, int wave, int lane, float partial) {
    float* st = reinterpret_cast<float*>(smem + BF_LDS_STAT) + slot * 128;
    partial += __shfl_xor(partial, 32, 64);
    if (lane < 32) st[wave * 32 + lane] = partial;
    __syncthreads();
    const int m = lane & 31;
    return st[m] + st[32 + m] + st[64 + m] + st[96 + m];
}

// [32][256] bf16 tile in LDS: 16-byte chunk ch (0..31) of row m lives at chunk ch ^ (m & 15)
__device__ __forceinline__ int bf_tile_off(int m, int col) { return m * 512 + ((((col >> 3) ^ (m & 15))) << 4) + ((col & 7) << 1); }

__device__ __forceinline__ void bf_tile_read(const char* smem, int lane, bf16x8 (&f)[BF_KS]) {
    const int m = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < BF_KS; ++ks)
        f[ks] = *reinterpret_cast<const bf16x8*>(smem + BF_LDS_TILE + m * 512 + (((2 * ks + h) ^ (m & 15)) << 4));
}

// accumulator initialised with the 32 bias values of hidden tile ht (register q of lane half h = feature (q&3) + 8 (q>>2) + 4 h)
__device__ __forceinline__ f32x16 bf_bias_init(const float* b1s, int ht, int h) {
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(b1s + 32 * ht + 8 * g + 4 * h);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[4 * g + i] = bb[i];
    }
    return acc;
}


template <typename K>
static int bf_prepare(K kernel, int lds_bytes = BF_LDS_TOTAL) {
    // (cheap and idempotent; called once per instantiation)
    if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) {
        cvft_set_error("block kernels: cannot reserve %d bytes of LDS", lds_bytes);
        return -2;
    }
    return 0;
}

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
