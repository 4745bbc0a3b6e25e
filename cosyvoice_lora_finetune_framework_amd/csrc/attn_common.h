// attn_common.h -- shared pieces of the fused attention kernels (head_dim 64).
// LDS tiles are [64 rows][64 cols] with row stride LDK.  Every MFMA operand is read
// through one of two loaders:
//   ld_kc : operand stored with its reduction index contiguous  ([rc][k])  -> 16-byte read
//   ld_km : operand stored reduction-index-major               ([k][rc])  -> strided gather
// (MFMA lane maps: cdna_hip_programming.md section 3.)
#pragma once
#include "common.h"

template <typename T> struct AttnCfg {
    static constexpr int VEC = 16 / sizeof(T);
    static constexpr int LDK = 64 + VEC;         // padded row stride (elements)
    static constexpr int CPR = 64 / VEC;         // 16-byte chunks per 64-wide row
    static constexpr int NK = 64 / Mma<T>::K;    // MFMA k-steps across 64
    static constexpr int TILE = 64 * LDK;        // elements per staged tile
};

// ---- kernel parameter block shared by every attention kernel (generic T kernels: attention.hip; bf16 32x32 MFMA
// kernels: attn_mfma32.hip)
template <typename T>
struct AP {
    int B, H, L;
    const T *q, *k, *v; int ld;
    const T* p; int ldp;
    const float *bu, *bv;
    const int* len;
    int causal; float scale;
    T* o; int ldo; float* lse;
    // bf16 kernels (attn_mfma32.hip): o_lo = bf16(O - bf16(O)), the part of the fp32 output the bf16 store drops, written by the
    // forward and read by the backward for delta = rowsum(dO . (O + O_lo)) -- with delta taken from the bf16-ROUNDED O alone the
    // score gradient dS = P (dP - delta) carried a common-mode error per query row that dominated dQ / dK wherever dP is nearly
    // constant over the keys (flat softmax over value rows that share a large mean: the estimator's mid blocks; 3 % .. 12 % on
    // dQ / dK against 0.17 % with the residual, tools/delta_error_model.py).  Same pitch as o; null = off.
    T* o_lo;
    const T* d_o; const float* delta;
    T *dq, *dk, *dv; int ldg;
    // attention-probability dropout (attention.py:118 `self.dropout(attn)`; DROP instantiations only)
    float drop_p; const long long* seed; unsigned site;
    int iso;      // REL = false: prompt-isolation split (modules.py:844-879); 0 = off
    float* dpos; int lddpos;      // REL backward: fp32 gradient w.r.t. p [2L-1, lddpos] (accumulated with atomics), or null
};

// keep-scale of score (b, h, i, j): 1/(1-p) or 0.  Counter-based (SplitMix64 finaliser, same as cvft_dropout_add), so
// forward and both backward kernels re-derive the same mask from (*seed, site, element index).
__device__ __forceinline__ float attn_keep_scale(unsigned long long key, unsigned long long idx, unsigned thr, float inv) {
    unsigned long long z = key + idx + 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    z ^= z >> 31;
    return (unsigned)z >= thr ? inv : 0.f;
}
__device__ __forceinline__ unsigned long long attn_drop_key(const long long* seed, unsigned site) {
    unsigned long long z = (unsigned long long)seed[0] ^ ((unsigned long long)site << 32);
    z += 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

// global rows r0..r0+63 (64 columns each, leading dim ld) -> LDS S[r][c]; rows outside [0,rlim) are zero
template <typename T>
__device__ __forceinline__ void stage64(const T* __restrict__ g, int ld, int r0, int rlim, T* S, int tid) {
    typedef AttnCfg<T> A;
    for (int c = tid; c < 64 * A::CPR; c += 256) {
        int r = c / A::CPR, cc = c % A::CPR;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r0 + r < rlim && r0 + r >= 0) v = *reinterpret_cast<const uint4*>(g + (size_t)(r0 + r) * ld + cc * A::VEC);
        *reinterpret_cast<uint4*>(&S[r * A::LDK + cc * A::VEC]) = v;
    }
}

// transposed staging: global rows r0..r0+63 -> LDS St[c][coff + r] (row stride ldt); optional straight copy S[r][c]
template <typename T>
__device__ __forceinline__ void stage64_tr(const T* __restrict__ g, int ld, int r0, int rlim, T* S, T* St, int ldt, int coff,
                                           int tid) {
    typedef AttnCfg<T> A;
    for (int c = tid; c < 64 * A::CPR; c += 256) {
        int r = c / A::CPR, cc = c % A::CPR;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r0 + r < rlim && r0 + r >= 0) v = *reinterpret_cast<const uint4*>(g + (size_t)(r0 + r) * ld + cc * A::VEC);
        if (S) *reinterpret_cast<uint4*>(&S[r * A::LDK + cc * A::VEC]) = v;
        const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
        for (int j = 0; j < A::VEC; ++j) St[(cc * A::VEC + j) * ldt + coff + r] = e[j];
    }
}

// ---- split staging (issue-early / write-late, cdna_hip_programming.md T14): the next tile's global loads are issued
// right after the barrier that publishes the current tile and stay in flight under its MFMAs; they are written to
// LDS at the top of the next iteration.  One 64x64 tile = 64*CPR 16-byte chunks over 256 threads.
template <typename T> struct TileRegs { uint4 v[(64 * AttnCfg<T>::CPR) / 256]; };

template <typename T>
__device__ __forceinline__ void tile_load(TileRegs<T>& t, const T* __restrict__ g, int ld, int r0, int rlim, int tid) {
    typedef AttnCfg<T> A;
#pragma unroll
    for (int i = 0; i < (64 * A::CPR) / 256; ++i) {
        const int c = tid + i * 256, r = c / A::CPR, cc = c % A::CPR;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r0 + r < rlim && r0 + r >= 0) v = *reinterpret_cast<const uint4*>(g + (size_t)(r0 + r) * ld + cc * A::VEC);
        t.v[i] = v;
    }
}
template <typename T>
__device__ __forceinline__ void tile_store(const TileRegs<T>& t, T* S, int tid) {
    typedef AttnCfg<T> A;
#pragma unroll
    for (int i = 0; i < (64 * A::CPR) / 256; ++i) {
        const int c = tid + i * 256, r = c / A::CPR, cc = c % A::CPR;
        *reinterpret_cast<uint4*>(&S[r * A::LDK + cc * A::VEC]) = t.v[i];
    }
}
// transposed image St[c][coff + r] (+ optional straight copy S[r][c])
template <typename T>
__device__ __forceinline__ void tile_store_tr(const TileRegs<T>& t, T* S, T* St, int ldt, int coff, int tid) {
    typedef AttnCfg<T> A;
#pragma unroll
    for (int i = 0; i < (64 * A::CPR) / 256; ++i) {
        const int c = tid + i * 256, r = c / A::CPR, cc = c % A::CPR;
        if (S) *reinterpret_cast<uint4*>(&S[r * A::LDK + cc * A::VEC]) = t.v[i];
        const T* e = reinterpret_cast<const T*>(&t.v[i]);
#pragma unroll
        for (int j = 0; j < A::VEC; ++j) St[(cc * A::VEC + j) * ldt + coff + r] = e[j];
    }
}

// ---- operand loaders -------------------------------------------------------------
template <typename T, typename S> struct FragLd;
template <> struct FragLd<bf16_t, bf16_t> {
    static __device__ __forceinline__ bf16x8 kc(const bf16_t* base, int ld, int rc0, int k0, int lane) {
        return *reinterpret_cast<const bf16x8*>(base + (rc0 + (lane & 15)) * ld + k0 + 8 * (lane >> 4));
    }
    // operand stored reduction-index-major ([k][rc], the natural row-major image of V / K / Q / dO): gfx950's
    // transposing LDS read (ds_read_b64_tr_b16, cdna_hip_programming.md T10) returns, per 16-lane group, a 4-row x
    // 16-column block column-major -- lane 4q+p supplies row q / columns 4p..4p+3, lane i receives column i.  Two reads
    // (rows 8g..8g+3 and 8g+4..8g+7) give the 8 consecutive k the MFMA operand wants; no transposed staging copy.
    // Needs EXEC all ones (every call site is wave-uniform) and 8-byte aligned addresses (ld % 4 == 0, rc0 % 4 == 0).
    static __device__ __forceinline__ bf16x8 km(const bf16_t* base, int ld, int rc0, int k0, int lane) {
        typedef __attribute__((address_space(3))) bf16x4 lds_b4;
        const bf16_t* p = base + (k0 + 8 * (lane >> 4) + ((lane >> 2) & 3)) * ld + rc0 + 4 * (lane & 3);
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)p);
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(p + 4 * ld));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
};
template <> struct FragLd<bf16_t, float> {
    static __device__ __forceinline__ bf16x8 kc(const float* base, int ld, int rc0, int k0, int lane) {
        bf16x8 f;
        const float* p = base + (rc0 + (lane & 15)) * ld + k0 + 8 * (lane >> 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (bf16_t)p[j];
        return f;
    }
};
// A[row = rc0 + (lane&15)][k] from an fp32 image stored [k][rc]; k >= klim reads as 0 (short reduction, padded k-step)
__device__ __forceinline__ bf16x8 frag_km_f32(const float* base, int ld, int rc0, int k0, int klim, int lane, bf16_t) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k0 + 8 * (lane >> 4) + j;
        f[j] = (k < klim) ? (bf16_t)base[k * ld + rc0 + (lane & 15)] : (bf16_t)0.f;
    }
    return f;
}
__device__ __forceinline__ float frag_km_f32(const float* base, int ld, int rc0, int k0, int klim, int lane, float) {
    const int k = k0 + (lane >> 4);
    return (k < klim) ? base[k * ld + rc0 + (lane & 15)] : 0.f;
}
template <> struct FragLd<float, float> {
    static __device__ __forceinline__ float kc(const float* base, int ld, int rc0, int k0, int lane) {
        return base[(rc0 + (lane & 15)) * ld + k0 + (lane >> 4)];
    }
    static __device__ __forceinline__ float km(const float* base, int ld, int rc0, int k0, int lane) {
        return base[(k0 + (lane >> 4)) * ld + rc0 + (lane & 15)];
    }
};

// add a per-column fp32 bias (indexed by the k position the fragment element holds)
__device__ __forceinline__ bf16x8 frag_add_bias(bf16x8 f, const float* bias, int k0, int lane) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (bf16_t)((float)f[j] + bias[k0 + 8 * (lane >> 4) + j]);
    return r;
}
__device__ __forceinline__ float frag_add_bias(float f, const float* bias, int k0, int lane) {
    return f + bias[k0 + (lane >> 4)];
}

// acc[nt] += Areg(16 x 64) . Btile(rows rc0 + nt*16 .., k-contiguous)^T
template <typename T, int NT>
__device__ __forceinline__ void mma_regA_kc(f32x4 (&acc)[NT], const typename Mma<T>::Frag (&a)[AttnCfg<T>::NK],
                                            const T* Bt, int ldb, int rc0, int lane) {
#pragma unroll
    for (int ks = 0; ks < AttnCfg<T>::NK; ++ks)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            Mma<T>::mma(acc[nt], a[ks], FragLd<T, T>::kc(Bt, ldb, rc0 + nt * 16, ks * Mma<T>::K, lane));
}

// reductions across the 16 lanes that share (lane>>4): one score-tile row lives on them
__device__ __forceinline__ float row16_max(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float row16_sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// (delta[b,h,i] = sum_d dO[i,d] * O[i,d] is formed inside the dQ kernel: attention.hip, attn_bwd_dq_body)
