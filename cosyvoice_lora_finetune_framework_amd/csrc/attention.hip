// attention.hip -- fused (flash-style) attention forward / backward, head_dim 64.
//   REL=false : softmax(QK^T*scale + (-1e10 on keys >= klen)) V      (U-Net estimator blocks)
//   REL=true  : softmax(((q+u)K^T + shift((q+v)P^T))*scale, -inf mask: pad / causal) V
// No (B,H,L,L) tensor is ever materialised: scores live in MFMA accumulators, the rel-pos
// term is produced per tile from an LDS-staged band of P rows and skew-read (rel_shift
// index law  bd[i,j] = (q_i+v).p[L-1-i+j]).  Backward = dQ kernel + dK/dV kernel, both
// recomputing P from the saved row log-sum-exp (no atomics, deterministic).
//
// Replaces (reference): modules.py:253-293 / diffusers Attention; attention.py:200-330, 82-127.
#include <stdlib.h>
#include "attn_common.h"

#define NEG_INF (-__builtin_inff())
// The per-wave LDS regions (Gw skew buffer, P / dS tiles) are written and read by the SAME wavefront: a wave's LDS
// operations complete in order, so draining its own LDS queue is enough -- no workgroup barrier (the rel-pos dK/dV
// kernel had ten s_barrier per tile, two of them real).
#define WAVE_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// ------------------------------------------------------------------------ forward
template <typename T, bool REL, bool DROP = false>
__global__ void __launch_bounds__(256) attn_fwd_kernel(AP<T> p) {
    typedef AttnCfg<T> A;
    typedef Mma<T> MM;
    typedef typename MM::Frag Frag;
    constexpr int LDK = A::LDK, NK = A::NK, LDG = 84;
    constexpr bool TR = false;                // (transposed staging copies retired: bf16 k-major operands use the transposing LDS read)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* Ks = reinterpret_cast<T*>(smem);
    T* Vs = Ks + A::TILE;
    T* Ps = Vs + A::TILE;                 // 4 x 16 x LDK
    T* Pb = Ps + 4 * 16 * LDK;            // REL: 128 x LDK
    float* Gs = reinterpret_cast<float*>(Pb + 128 * LDK);   // REL: 4 x 16 x LDG

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int q0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const int L = p.L;
    const size_t rowbase = (size_t)b * L;
    const T* qg = p.q + rowbase * p.ld + h * 64;
    const T* kg = p.k + rowbase * p.ld + h * 64;
    const T* vg = p.v + rowbase * p.ld + h * 64;
    const T* pg = REL ? p.p + h * 64 : nullptr;
    const int lb = p.len ? p.len[b] : L;
    unsigned long long dkey = 0;
    unsigned dthr = 0;
    float dinv = 1.f;
    if (DROP) {
        dkey = attn_drop_key(p.seed, p.site);
        dthr = (unsigned)fminf(4294967295.f, p.drop_p * 4294967296.f);
        dinv = 1.f / (1.f - p.drop_p);
    }

    int jmax;
    if (REL) {
        jmax = min(L, lb);
        if (p.causal) jmax = min(jmax, q0 + 64);
    } else {
        jmax = (lb >= 1) ? min(L, lb) : L;
    }
    TileRegs<T> kr, vr, pr0, pr1;
    auto prefetch = [&](int j0) __attribute__((always_inline)) {
        tile_load(kr, kg, p.ld, j0, L, tid);
        tile_load(vr, vg, p.ld, j0, L, tid);
        if (REL) {
            const int mb = (L - 1) - (q0 + 63) + j0;
            tile_load(pr0, pg, p.ldp, mb, 2 * L - 1, tid);
            tile_load(pr1, pg, p.ldp, mb + 64, 2 * L - 1, tid);
        }
    };
    if (jmax > 0) prefetch(0);

    stage64(qg, p.ld, q0, L, Ks, tid);
    __syncthreads();
    Frag qf[NK], qv[NK];
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        qf[ks] = FragLd<T, T>::kc(Ks, LDK, 16 * w, ks * MM::K, lane);
        qv[ks] = qf[ks];
        if (REL) {
            qv[ks] = frag_add_bias(qf[ks], p.bv + h * 64, ks * MM::K, lane);
            qf[ks] = frag_add_bias(qf[ks], p.bu + h * 64, ks * MM::K, lane);
        }
    }
    __syncthreads();

    float m_run[4], l_run[4];
    f32x4 oacc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { m_run[r] = NEG_INF; l_run[r] = 0.f; oacc[r] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    T* Pw = Ps + w * 16 * LDK;
    float* Gw = Gs + w * 16 * LDG;

    for (int j0 = 0; j0 < jmax; j0 += 64) {
        tile_store(kr, Ks, tid);
        if (TR) tile_store_tr<T>(vr, nullptr, Vs, LDK, 0, tid);      // bf16: V only ever read k(=kv)-major => keep V^T
        else tile_store(vr, Vs, tid);
        if (REL) {
            tile_store(pr0, Pb, tid);
            tile_store(pr1, Pb + 64 * LDK, tid);
        }
        __syncthreads();
        if (j0 + 64 < jmax) prefetch(j0 + 64);                       // in flight under this tile's MFMAs
        f32x4 s[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) s[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        mma_regA_kc<T, 4>(s, qf, Ks, LDK, 0, lane);
        if (REL) {
            f32x4 g[5];
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) g[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            mma_regA_kc<T, 5>(g, qv, Pb, LDK, 48 - 16 * w, lane);
#pragma unroll
            for (int nt = 0; nt < 5; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) Gw[((lane >> 4) * 4 + r) * LDG + nt * 16 + (lane & 15)] = g[nt][r];
            WAVE_LDS_SYNC();
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int row = (lane >> 4) * 4 + r;
                    s[nt][r] += Gw[row * LDG + 15 - row + nt * 16 + (lane & 15)];
                }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = q0 + 16 * w + (lane >> 4) * 4 + r;
            float tm = NEG_INF;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int j = j0 + nt * 16 + (lane & 15);
                float x = s[nt][r] * p.scale;
                if (REL) {
                    bool valid = (j < lb) && (j < L) && (!p.causal || j <= i);
                    x = valid ? x : NEG_INF;
                } else {
                    x = (j < L) ? x + (j < lb ? 0.f : -1.0e10f) : NEG_INF;
                    if (p.iso > 0 && ((i < p.iso) != (j < p.iso))) x = NEG_INF;      // prompt / target segments do not mix
                }
                s[nt][r] = x;
                tm = fmaxf(tm, x);
            }
            tm = row16_max(tm);
            const float mn = fmaxf(m_run[r], tm);
            const float ms = (mn == NEG_INF) ? 0.f : mn;
            const float alpha = __expf(m_run[r] - ms);
            float rs = 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                float pv = __expf(s[nt][r] - ms);
                s[nt][r] = pv;
                rs += pv;
            }
            rs = row16_sum(rs);
            l_run[r] = l_run[r] * alpha + rs;
            m_run[r] = mn;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) oacc[dt][r] *= alpha;
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float pv = s[nt][r];
                if (DROP) {     // the softmax denominator keeps the undropped sum; only the PV operand is masked
                    const unsigned long long i = q0 + 16 * w + (lane >> 4) * 4 + r, j = j0 + nt * 16 + (lane & 15);
                    pv *= attn_keep_scale(dkey, (((unsigned long long)b * p.H + h) * L + i) * L + j, dthr, dinv);
                }
                Pw[((lane >> 4) * 4 + r) * LDK + nt * 16 + (lane & 15)] = from_f32<T>(pv);
            }
        WAVE_LDS_SYNC();
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            Frag a = FragLd<T, T>::kc(Pw, LDK, 0, ks * MM::K, lane);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                MM::mma(oacc[dt], a, TR ? FragLd<T, T>::kc(Vs, LDK, dt * 16, ks * MM::K, lane)
                                        : FragLd<T, T>::km(Vs, LDK, dt * 16, ks * MM::K, lane));
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = q0 + 16 * w + (lane >> 4) * 4 + r;
        if (i >= L) continue;
        const float inv = l_run[r] > 0.f ? 1.f / l_run[r] : 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            p.o[(rowbase + i) * p.ldo + h * 64 + dt * 16 + (lane & 15)] = from_f32<T>(oacc[dt][r] * inv);
        if ((lane & 15) == 0)
            p.lse[((size_t)b * p.H + h) * L + i] = l_run[r] > 0.f ? m_run[r] + logf(l_run[r]) : __builtin_inff();
    }
}

// ------------------------------------------------------------------------ backward: dQ
template <typename T, bool REL, bool DROP = false, bool DPOS = false>
__device__ __forceinline__ void attn_bwd_dq_body(const AP<T>& p, const int bx) {
    typedef AttnCfg<T> A;
    typedef Mma<T> MM;
    typedef typename MM::Frag Frag;
    constexpr int LDK = A::LDK, NK = A::NK, LDG = 100;
    constexpr bool TR = false;
    constexpr int KB = (sizeof(T) == 2) ? 96 : 80;     // skewed dS width, padded to the MFMA k-step
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* Ks = reinterpret_cast<T*>(smem);
    T* Vs = Ks + A::TILE;
    T* Ds = Vs + A::TILE;                 // 4 x 16 x LDK
    T* Pb = Ds + 4 * 16 * LDK;            // REL: 192 x LDK
    float* Gs = reinterpret_cast<float*>(Pb + (REL ? 192 * LDK : 0));   // REL: 4 x 16 x LDG
    T* Kt = reinterpret_cast<T*>(Gs + (REL ? 4 * 16 * LDG : 0));        // TR: K^T tile [d][kv]
    // DPOS (gradient w.r.t. p, i.e. LoRA on linear_pos -- lora.py:155-166 default targets): the wave's 16 rows of q + v,
    // [i][d], padded with zero rows to the MFMA k-step; B operand of  dP[m,:] += sum_i dS_skewed[i,m] (q_i + v)
    constexpr int QVR = (sizeof(T) == 2) ? 32 : 16;
    T* QVs = Kt;                          // 4 x QVR x LDK   (TR is retired: Kt has no other user)

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int q0 = bx * 64, h = blockIdx.y, b = blockIdx.z;
    const int L = p.L;
    const size_t rowbase = (size_t)b * L;
    const T* qg = p.q + rowbase * p.ld + h * 64;
    const T* kg = p.k + rowbase * p.ld + h * 64;
    const T* vg = p.v + rowbase * p.ld + h * 64;
    const T* dog = p.d_o + rowbase * p.ldo + h * 64;
    const T* pg = REL ? p.p + h * 64 : nullptr;
    const int lb = p.len ? p.len[b] : L;
    unsigned long long dkey = 0;
    unsigned dthr = 0;
    float dinv = 1.f;
    if (DROP) {
        dkey = attn_drop_key(p.seed, p.site);
        dthr = (unsigned)fminf(4294967295.f, p.drop_p * 4294967296.f);
        dinv = 1.f / (1.f - p.drop_p);
    }

    int jmax;
    if (REL) {
        jmax = min(L, lb);
        if (p.causal) jmax = min(jmax, q0 + 64);
    } else {
        jmax = (lb >= 1) ? min(L, lb) : L;
    }
    TileRegs<T> kr, vr, pr0, pr1, pr2;
    auto prefetch = [&](int j0) __attribute__((always_inline)) {
        tile_load(kr, kg, p.ld, j0, L, tid);
        tile_load(vr, vg, p.ld, j0, L, tid);
        if (REL) {
            const int mb = (L - 1) - (q0 + 63) + j0;
            tile_load(pr0, pg, p.ldp, mb, 2 * L - 1, tid);
            tile_load(pr1, pg, p.ldp, mb + 64, 2 * L - 1, tid);
            tile_load(pr2, pg, p.ldp, mb + 128, 2 * L - 1, tid);
        }
    };
    if (jmax > 0) prefetch(0);

    stage64(qg, p.ld, q0, L, Ks, tid);
    stage64(dog, p.ldo, q0, L, Vs, tid);
    __syncthreads();
    Frag qf[NK], qv[NK], dof[NK];
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        qf[ks] = FragLd<T, T>::kc(Ks, LDK, 16 * w, ks * MM::K, lane);
        dof[ks] = FragLd<T, T>::kc(Vs, LDK, 16 * w, ks * MM::K, lane);
        qv[ks] = qf[ks];
        if (REL) {
            qv[ks] = frag_add_bias(qf[ks], p.bv + h * 64, ks * MM::K, lane);
            qf[ks] = frag_add_bias(qf[ks], p.bu + h * 64, ks * MM::K, lane);
        }
    }
    // delta[i] = sum_d dO[i][d] * O[i][d], computed here (it used to be a launch of its own, 90 per step): lane l of wave w
    // sums 16 columns of row 16w + l/4 (dO from the staged tile, O from global), 4-lane reduce; the rows a lane needs
    // come by shuffle, and the values are published for the dK/dV kernel, which runs after this one on the stream.
    float dsum = 0.f;
    {
        const int row = 16 * w + (lane >> 2), c0 = (lane & 3) * 16, i = q0 + row;
        if (i < L) {
            const T* orow = p.o + (rowbase + i) * p.ldo + h * 64 + c0;
            const T* drow = Vs + row * LDK + c0;
            constexpr int NV = 16 / A::VEC;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const uint4 ov = *reinterpret_cast<const uint4*>(orow + v * A::VEC);
                const uint4 dv = *reinterpret_cast<const uint4*>(drow + v * A::VEC);
                const T* oe = reinterpret_cast<const T*>(&ov);
                const T* de = reinterpret_cast<const T*>(&dv);
#pragma unroll
                for (int e = 0; e < A::VEC; ++e) dsum += to_f32(oe[e]) * to_f32(de[e]);
            }
        }
        dsum += __shfl_xor(dsum, 1);
        dsum += __shfl_xor(dsum, 2);
        if ((lane & 3) == 0 && i < L) const_cast<float*>(p.delta)[((size_t)b * p.H + h) * L + i] = dsum;
    }
    float lse_r[4], del_r[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = q0 + 16 * w + (lane >> 4) * 4 + r;
        lse_r[r] = (i < L) ? p.lse[((size_t)b * p.H + h) * L + i] : __builtin_inff();
        del_r[r] = __shfl(dsum, 4 * ((lane >> 4) * 4 + r));
    }
    __syncthreads();

    f32x4 dqacc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) dqacc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    T* Dw = Ds + w * 16 * LDK;
    float* Gw = Gs + w * 16 * LDG;
    T* QVw = QVs + w * QVR * LDK;
    if (REL && DPOS) {
        for (int e = lane; e < QVR * 64; e += 64) {
            const int row = e >> 6, c = e & 63, i = q0 + 16 * w + row;
            float v = 0.f;
            if (row < 16 && i < L) v = to_f32(qg[(size_t)i * p.ld + c]) + p.bv[h * 64 + c];
            QVw[row * LDK + c] = from_f32<T>(v);
        }
        WAVE_LDS_SYNC();
    }

    for (int j0 = 0; j0 < jmax; j0 += 64) {
        if (TR) tile_store_tr<T>(kr, Ks, Kt, LDK, 0, tid);
        else tile_store(kr, Ks, tid);
        tile_store(vr, Vs, tid);
        if (REL) {
            // (a transposed copy of the band was measured slower than strided reads here: 3 scatter passes per tile)
            tile_store(pr0, Pb, tid);
            tile_store(pr1, Pb + 64 * LDK, tid);
            tile_store(pr2, Pb + 128 * LDK, tid);
        }
        __syncthreads();
        if (j0 + 64 < jmax) prefetch(j0 + 64);
        f32x4 s[4], dp[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { s[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        mma_regA_kc<T, 4>(s, qf, Ks, LDK, 0, lane);
        mma_regA_kc<T, 4>(dp, dof, Vs, LDK, 0, lane);
        if (REL) {
            f32x4 g[5];
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) g[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            mma_regA_kc<T, 5>(g, qv, Pb, LDK, 48 - 16 * w, lane);
#pragma unroll
            for (int nt = 0; nt < 5; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) Gw[((lane >> 4) * 4 + r) * LDG + nt * 16 + (lane & 15)] = g[nt][r];
            WAVE_LDS_SYNC();
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int row = (lane >> 4) * 4 + r;
                    s[nt][r] += Gw[row * LDG + 15 - row + nt * 16 + (lane & 15)];
                }
            WAVE_LDS_SYNC();
            for (int e = lane; e < 16 * KB; e += 64) Gw[(e / KB) * LDG + (e % KB)] = 0.f;
            WAVE_LDS_SYNC();
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = q0 + 16 * w + (lane >> 4) * 4 + r;
            const int row = (lane >> 4) * 4 + r;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int j = j0 + nt * 16 + (lane & 15);
                float x = s[nt][r] * p.scale;
                bool valid;
                if (REL) {
                    valid = (j < lb) && (j < L) && (!p.causal || j <= i);
                } else {
                    valid = (j < L) && !(p.iso > 0 && ((i < p.iso) != (j < p.iso)));
                    x += (j < lb ? 0.f : -1.0e10f);
                }
                const float pv = valid ? __expf(x - lse_r[r]) : 0.f;
                float dpe = dp[nt][r];
                if (DROP) dpe *= attn_keep_scale(dkey, (((unsigned long long)b * p.H + h) * L + i) * L + j, dthr, dinv);
                const float ds = pv * (dpe - del_r[r]) * p.scale;
                Dw[row * LDK + nt * 16 + (lane & 15)] = from_f32<T>(ds);
                if (REL) Gw[row * LDG + 15 - row + nt * 16 + (lane & 15)] = ds;
            }
        }
        WAVE_LDS_SYNC();
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            Frag a = FragLd<T, T>::kc(Dw, LDK, 0, ks * MM::K, lane);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                MM::mma(dqacc[dt], a, TR ? FragLd<T, T>::kc(Kt, LDK, dt * 16, ks * MM::K, lane)
                                         : FragLd<T, T>::km(Ks, LDK, dt * 16, ks * MM::K, lane));
        }
        if (REL) {
#pragma unroll
            for (int k0 = 0; k0 < KB; k0 += MM::K) {
                Frag a = FragLd<T, float>::kc(Gw, LDG, 0, k0, lane);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    MM::mma(dqacc[dt], a, FragLd<T, T>::km(Pb, LDK, dt * 16, (48 - 16 * w) + k0, lane));
            }
            if (DPOS) {
                // Gw column c is p row  mband + c;  dP[mband + c, :] += sum_i Gw[i][c] (q_i + v)   (fp32 atomics:
                // the sum runs over query blocks, heads' batches and key tiles)
                const int mband = (L - 1) - (q0 + 63) + j0 + (48 - 16 * w);
#pragma unroll 1
                for (int mt = 0; mt < KB / 16; ++mt) {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int k0 = 0; k0 < QVR; k0 += MM::K)
                            MM::mma(acc, frag_km_f32(Gw, LDG, mt * 16, k0, 16, lane, T()), FragLd<T, T>::km(QVw, LDK, dt * 16, k0, lane));
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int m = mband + mt * 16 + (lane >> 4) * 4 + r;
                            if (m >= 0 && m <= 2 * L - 2)
                                atomicAdd(p.dpos + (size_t)m * p.lddpos + h * 64 + dt * 16 + (lane & 15), acc[r]);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = q0 + 16 * w + (lane >> 4) * 4 + r;
        if (i >= L) continue;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            p.dq[(rowbase + i) * p.ldg + h * 64 + dt * 16 + (lane & 15)] = from_f32<T>(dqacc[dt][r]);
    }
}

// ------------------------------------------------------------------------ backward: dK, dV
template <typename T, bool REL, bool DROP = false, bool DPOS = false>
__global__ void __launch_bounds__(256) attn_bwd_dq_kernel(AP<T> p) { attn_bwd_dq_body<T, REL, DROP, DPOS>(p, blockIdx.x); }

template <typename T, bool REL, bool DROP = false>
__device__ __forceinline__ void attn_bwd_dkv_body(const AP<T>& p, const int bx) {
    typedef AttnCfg<T> A;
    typedef Mma<T> MM;
    typedef typename MM::Frag Frag;
    constexpr int LDK = A::LDK, NK = A::NK, LDG = 36;
    constexpr bool TR = false;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* Ks = reinterpret_cast<T*>(smem);
    T* Vs = Ks + A::TILE;
    T* Qs = Vs + A::TILE;
    T* Os = Qs + A::TILE;
    T* Pt = Os + A::TILE;                 // 4 x 16 x LDK  : P^T  [j][i]
    T* Dt = Pt + 4 * 16 * LDK;            // 4 x 16 x LDK  : dS^T [j][i]
    float* lse_s = reinterpret_cast<float*>(Dt + 4 * 16 * LDK);   // 64
    float* del_s = lse_s + 64;                                     // 64
    T* Pb = reinterpret_cast<T*>(del_s + 64);                      // REL: 128 x LDK
    float* Gs = reinterpret_cast<float*>(Pb + (REL ? 128 * LDK : 0));          // REL: 4 x 16 x LDG
    T* Qt = reinterpret_cast<T*>(Gs + (REL ? 4 * 16 * LDG : 0));   // TR: Q^T  [d][i]
    T* Ot = Qt + A::TILE;                                          // TR: dO^T [d][i]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j0 = bx * 64, h = blockIdx.y, b = blockIdx.z;
    const int L = p.L;
    const size_t rowbase = (size_t)b * L;
    const T* qg = p.q + rowbase * p.ld + h * 64;
    const T* kg = p.k + rowbase * p.ld + h * 64;
    const T* vg = p.v + rowbase * p.ld + h * 64;
    const T* dog = p.d_o + rowbase * p.ldo + h * 64;
    const T* pg = REL ? p.p + h * 64 : nullptr;
    const int lb = p.len ? p.len[b] : L;
    unsigned long long dkey = 0;
    unsigned dthr = 0;
    float dinv = 1.f;
    if (DROP) {
        dkey = attn_drop_key(p.seed, p.site);
        dthr = (unsigned)fminf(4294967295.f, p.drop_p * 4294967296.f);
        dinv = 1.f / (1.f - p.drop_p);
    }
    const int lb_eff = REL ? min(L, lb) : ((lb >= 1) ? min(L, lb) : L);
    const int jw = j0 + 16 * w;

    f32x4 dkacc[4], dvacc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { dkacc[r] = f32x4{0.f, 0.f, 0.f, 0.f}; dvacc[r] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float csum = 0.f;

    if (j0 < lb_eff) {   // block-uniform
        stage64(kg, p.ld, j0, L, Ks, tid);
        stage64(vg, p.ld, j0, L, Vs, tid);
        __syncthreads();
        Frag kf[NK], vf[NK];
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            kf[ks] = FragLd<T, T>::kc(Ks, LDK, 16 * w, ks * MM::K, lane);
            vf[ks] = FragLd<T, T>::kc(Vs, LDK, 16 * w, ks * MM::K, lane);
        }
        T* Pw = Pt + w * 16 * LDK;
        T* Dw = Dt + w * 16 * LDK;
        float* Gw = Gs + w * 16 * LDG;
        const int ibeg = (REL && p.causal) ? j0 : 0;
        TileRegs<T> qr, dor, pr0, pr1;
        float lse_n = 0.f, del_n = 0.f;
        auto prefetch = [&](int i0) __attribute__((always_inline)) {
            tile_load(qr, qg, p.ld, i0, L, tid);
            tile_load(dor, dog, p.ldo, i0, L, tid);
            if (tid < 64) {
                const int i = i0 + tid;
                lse_n = (i < L) ? p.lse[((size_t)b * p.H + h) * L + i] : __builtin_inff();
                del_n = (i < L) ? p.delta[((size_t)b * p.H + h) * L + i] : 0.f;
            }
            if (REL) {
                const int mb = (L - 1) - (i0 + 63) + j0;
                tile_load(pr0, pg, p.ldp, mb, 2 * L - 1, tid);
                tile_load(pr1, pg, p.ldp, mb + 64, 2 * L - 1, tid);
            }
        };
        if (ibeg < L) prefetch(ibeg);
        for (int i0 = ibeg; i0 < L; i0 += 64) {
            if (TR) {
                tile_store_tr<T>(qr, Qs, Qt, LDK, 0, tid);
                tile_store_tr<T>(dor, Os, Ot, LDK, 0, tid);
            } else {
                tile_store(qr, Qs, tid);
                tile_store(dor, Os, tid);
            }
            if (tid < 64) {
                lse_s[tid] = lse_n;
                del_s[tid] = del_n;
            }
            if (REL) {
                tile_store(pr0, Pb, tid);
                tile_store(pr1, Pb + 64 * LDK, tid);
            }
            __syncthreads();
            if (i0 + 64 < L) prefetch(i0 + 64);
#pragma unroll 1
            for (int mt = 0; mt < 4; ++mt) {
                Frag au[NK], av[NK], ad[NK];
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) {
                    au[ks] = FragLd<T, T>::kc(Qs, LDK, mt * 16, ks * MM::K, lane);
                    ad[ks] = FragLd<T, T>::kc(Os, LDK, mt * 16, ks * MM::K, lane);
                    av[ks] = au[ks];
                    if (REL) {
                        av[ks] = frag_add_bias(au[ks], p.bv + h * 64, ks * MM::K, lane);
                        au[ks] = frag_add_bias(au[ks], p.bu + h * 64, ks * MM::K, lane);
                    }
                }
                f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) {
                    MM::mma(s, au[ks], kf[ks]);
                    MM::mma(dp, ad[ks], vf[ks]);
                }
                if (REL) {
                    f32x4 g[2];
                    g[0] = f32x4{0.f, 0.f, 0.f, 0.f};
                    g[1] = f32x4{0.f, 0.f, 0.f, 0.f};
                    mma_regA_kc<T, 2>(g, av, Pb, LDK, (48 - 16 * mt) + 16 * w, lane);
                    WAVE_LDS_SYNC();   // previous mt's skew reads done
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) Gw[((lane >> 4) * 4 + r) * LDG + nt * 16 + (lane & 15)] = g[nt][r];
                    WAVE_LDS_SYNC();
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int row = (lane >> 4) * 4 + r;
                        s[r] += Gw[row * LDG + 15 - row + (lane & 15)];
                    }
                }
                const int j = jw + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int il = mt * 16 + (lane >> 4) * 4 + r;
                    const int i = i0 + il;
                    float x = s[r] * p.scale;
                    bool valid;
                    if (REL) {
                        valid = (j < lb) && (j < L) && (i < L) && (!p.causal || j <= i);
                    } else {
                        valid = (j < L) && (i < L) && !(p.iso > 0 && ((i < p.iso) != (j < p.iso)));
                        x += (j < lb ? 0.f : -1.0e10f);
                    }
                    const float pv = valid ? __expf(x - lse_s[il]) : 0.f;
                    float ks = 1.f;
                    if (DROP) ks = attn_keep_scale(dkey, (((unsigned long long)b * p.H + h) * L + i) * L + j, dthr, dinv);
                    const float ds = pv * (dp[r] * ks - del_s[il]) * p.scale;
                    Pw[(lane & 15) * LDK + il] = from_f32<T>(pv * ks);
                    Dw[(lane & 15) * LDK + il] = from_f32<T>(ds);
                    csum += ds;
                }
            }
            WAVE_LDS_SYNC();
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                Frag ap = FragLd<T, T>::kc(Pw, LDK, 0, ks * MM::K, lane);
                Frag ads = FragLd<T, T>::kc(Dw, LDK, 0, ks * MM::K, lane);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    MM::mma(dvacc[dt], ap, TR ? FragLd<T, T>::kc(Ot, LDK, dt * 16, ks * MM::K, lane)
                                              : FragLd<T, T>::km(Os, LDK, dt * 16, ks * MM::K, lane));
                    MM::mma(dkacc[dt], ads, TR ? FragLd<T, T>::kc(Qt, LDK, dt * 16, ks * MM::K, lane)
                                               : FragLd<T, T>::km(Qs, LDK, dt * 16, ks * MM::K, lane));
                }
            }
            __syncthreads();
        }
        if (REL) {   // dK_j += (sum_i dS[i,j]) * u   (A operand was raw q)
            float ct = csum + __shfl_xor(csum, 16, 64);
            ct += __shfl_xor(ct, 32, 64);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float cj = __shfl(ct, (lane >> 4) * 4 + r, 64);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) dkacc[dt][r] += cj * p.bu[h * 64 + dt * 16 + (lane & 15)];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int j = jw + (lane >> 4) * 4 + r;
        if (j >= L) continue;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const size_t off = (rowbase + j) * p.ldg + h * 64 + dt * 16 + (lane & 15);
            p.dk[off] = from_f32<T>(dkacc[dt][r]);
            p.dv[off] = from_f32<T>(dvacc[dt][r]);
        }
    }
}

template <typename T, bool REL, bool DROP = false>
__global__ void __launch_bounds__(256) attn_bwd_dkv_kernel(AP<T> p) { attn_bwd_dkv_body<T, REL, DROP>(p, blockIdx.x); }

// ------------------------------------------------------------------------ host side
// bf16 launches go to the 32x32 MFMA kernels of attn_mfma32.hip (CVFT_ATTN_V1=1 keeps the generic kernels below, which
// also serve fp32 and the gradient w.r.t. the projected positional encoding).
int cvft_attn32_fwd(const AP<bf16_t>& p, int rel, hipStream_t st);
int cvft_attn32_bwd(const AP<bf16_t>& p, int rel, hipStream_t st);
static bool attn_v1() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("CVFT_ATTN_V1");
        v = (e && e[0] == '1') ? 1 : 0;
    }
    return v == 1;
}
static bool al8(const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr) {
    return ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d)) & 7) == 0;
}
template <typename T> static size_t smem_fwd(bool rel) {
    typedef AttnCfg<T> A;
    size_t s = (size_t)(2 * A::TILE + 4 * 16 * A::LDK) * sizeof(T);
    if (rel) s += (size_t)128 * A::LDK * sizeof(T) + 4 * 16 * 84 * sizeof(float);
    return s;
}
template <typename T> static size_t smem_dq(bool rel, bool dpos = false) {
    typedef AttnCfg<T> A;
    size_t s = (size_t)(2 * A::TILE + 4 * 16 * A::LDK) * sizeof(T);
    if (rel) s += (size_t)192 * A::LDK * sizeof(T) + 4 * 16 * 100 * sizeof(float);
    if (dpos) s += (size_t)4 * (sizeof(T) == 2 ? 32 : 16) * A::LDK * sizeof(T);
    return s;
}
template <typename T> static size_t smem_dkv(bool rel) {
    typedef AttnCfg<T> A;
    size_t s = (size_t)(4 * A::TILE + 2 * 4 * 16 * A::LDK) * sizeof(T) + 128 * sizeof(float);
    if (rel) s += (size_t)128 * A::LDK * sizeof(T) + 4 * 16 * 36 * sizeof(float);
    return s;
}

template <typename K>
static int set_smem(K kernel, size_t bytes, const char* name) {
    static K seen[16];                        // kernels of this signature whose attribute is already set
    static int nseen = 0;
    for (int i = 0; i < nseen; ++i)
        if (seen[i] == kernel) return 0;
    if (nseen < 16) seen[nseen++] = kernel;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        cvft_set_error("%s: hipFuncSetAttribute(%zu bytes) failed: %s", name, bytes, hipGetErrorString(e));
        return -2;
    }
    return 0;
}

template <typename T, bool REL>
static int launch_fwd(const AP<T>& p, hipStream_t st) {
    size_t sm = smem_fwd<T>(REL);
    dim3 grid((p.L + 63) / 64, p.H, p.B);
    if constexpr (REL) {
        if (p.drop_p > 0.f) {
            if (set_smem(attn_fwd_kernel<T, true, true>, sm, "attn_fwd")) return -2;
            hipLaunchKernelGGL((attn_fwd_kernel<T, true, true>), grid, dim3(256), sm, st, p);
            CVFT_LAUNCH_CHECK("attn_fwd");
            return 0;
        }
    }
    if (set_smem(attn_fwd_kernel<T, REL>, sm, "attn_fwd")) return -2;
    hipLaunchKernelGGL((attn_fwd_kernel<T, REL>), grid, dim3(256), sm, st, p);
    CVFT_LAUNCH_CHECK("attn_fwd");
    return 0;
}
template <typename T, bool REL>
static int launch_bwd(const AP<T>& p_in, float* delta, const T* o, hipStream_t st) {
    AP<T> p = p_in;
    p.o = const_cast<T*>(o);              // read-only here: the dQ kernel forms delta = rowsum(dO * O) itself
    (void)delta;
    size_t s1 = smem_dq<T>(REL), s2 = smem_dkv<T>(REL);
    // (one merged launch for both roles was measured slower than two launches: 36.8 vs 36.5 ms/step)
    dim3 grid((p.L + 63) / 64, p.H, p.B);
    if constexpr (REL) {
        if (p.dpos) {      // gradient w.r.t. p requested: the dQ kernel also scatters dP
            const size_t s1p = smem_dq<T>(true, true);
            if (p.drop_p > 0.f) {
                if (set_smem(attn_bwd_dq_kernel<T, true, true, true>, s1p, "attn_bwd_dq")) return -2;
                if (set_smem(attn_bwd_dkv_kernel<T, true, true>, s2, "attn_bwd_dkv")) return -2;
                hipLaunchKernelGGL((attn_bwd_dq_kernel<T, true, true, true>), grid, dim3(256), s1p, st, p);
                CVFT_LAUNCH_CHECK("attn_bwd_dq");
                hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, true, true>), grid, dim3(256), s2, st, p);
                CVFT_LAUNCH_CHECK("attn_bwd_dkv");
                return 0;
            }
            if (set_smem(attn_bwd_dq_kernel<T, true, false, true>, s1p, "attn_bwd_dq")) return -2;
            if (set_smem(attn_bwd_dkv_kernel<T, true, false>, s2, "attn_bwd_dkv")) return -2;
            hipLaunchKernelGGL((attn_bwd_dq_kernel<T, true, false, true>), grid, dim3(256), s1p, st, p);
            CVFT_LAUNCH_CHECK("attn_bwd_dq");
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, true, false>), grid, dim3(256), s2, st, p);
            CVFT_LAUNCH_CHECK("attn_bwd_dkv");
            return 0;
        }
        if (p.drop_p > 0.f) {
            if (set_smem(attn_bwd_dq_kernel<T, true, true>, s1, "attn_bwd_dq")) return -2;
            if (set_smem(attn_bwd_dkv_kernel<T, true, true>, s2, "attn_bwd_dkv")) return -2;
            hipLaunchKernelGGL((attn_bwd_dq_kernel<T, true, true>), grid, dim3(256), s1, st, p);
            CVFT_LAUNCH_CHECK("attn_bwd_dq");
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, true, true>), grid, dim3(256), s2, st, p);
            CVFT_LAUNCH_CHECK("attn_bwd_dkv");
            return 0;
        }
    }
    if (set_smem(attn_bwd_dq_kernel<T, REL>, s1, "attn_bwd_dq")) return -2;
    if (set_smem(attn_bwd_dkv_kernel<T, REL>, s2, "attn_bwd_dkv")) return -2;
    hipLaunchKernelGGL((attn_bwd_dq_kernel<T, REL>), grid, dim3(256), s1, st, p);
    CVFT_LAUNCH_CHECK("attn_bwd_dq");
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, REL>), grid, dim3(256), s2, st, p);
    CVFT_LAUNCH_CHECK("attn_bwd_dkv");
    return 0;
}

static int check_common(const char* name, int dtype, int B, int H, int L, int ld, int ldo, const void* q, const void* k,
                        const void* v) {
    CVFT_CHECK_ARG(dtype == CVFT_F32 || dtype == CVFT_BF16, "%s: bad dtype", name);
    CVFT_CHECK_ARG(B > 0 && H > 0 && L > 0 && B <= 65535 && H <= 65535, "%s: bad dims B%d H%d L%d", name, B, H, L);
    int vec = dtype == CVFT_BF16 ? 8 : 4;
    CVFT_CHECK_ARG(ld >= H * 64 && ldo >= H * 64 && ld % vec == 0 && ldo % vec == 0, "%s: bad leading dims", name);
    CVFT_CHECK_ARG(q && k && v, "%s: null operand", name);
    CVFT_CHECK_ARG((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0, "%s: q/k/v must be 16-byte aligned", name);
    return 0;
}

template <typename T>
static AP<T> make_ap(int B, int H, int L, const void* q, const void* k, const void* v, int ld, const void* pp, int ldp,
                     const float* bu, const float* bv, const int32_t* len, int causal, float scale) {
    AP<T> a;
    a.B = B; a.H = H; a.L = L; a.q = (const T*)q; a.k = (const T*)k; a.v = (const T*)v; a.ld = ld;
    a.p = (const T*)pp; a.ldp = ldp; a.bu = bu; a.bv = bv; a.len = len; a.causal = causal; a.scale = scale;
    a.o = nullptr; a.o_lo = nullptr; a.ldo = 0; a.lse = nullptr; a.d_o = nullptr; a.delta = nullptr; a.dq = a.dk = a.dv = nullptr; a.ldg = 0;
    a.drop_p = 0.f; a.seed = nullptr; a.site = 0; a.iso = 0; a.dpos = nullptr; a.lddpos = 0;
    return a;
}

extern "C" int cvft_attn_bias_fwd(int dtype, int B, int H, int T_, const void* q, const void* k, const void* v, int ld,
                                  const int32_t* klen, float scale, int iso_len, void* o, int ldo, float* lse, void* o_lo, void* stream) {
    if (check_common("cvft_attn_bias_fwd", dtype, B, H, T_, ld, ldo, q, k, v)) return -1;
    CVFT_CHECK_ARG(o && lse, "cvft_attn_bias_fwd: null output");
    CVFT_CHECK_ARG(!o_lo || (dtype == CVFT_BF16 && (((uintptr_t)o_lo) & 7) == 0), "cvft_attn_bias_fwd: o_lo is a bf16 buffer (8-byte aligned) or NULL");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == CVFT_F32) {
        AP<float> a = make_ap<float>(B, H, T_, q, k, v, ld, nullptr, 0, nullptr, nullptr, klen, 0, scale);
        a.o = (float*)o; a.ldo = ldo; a.lse = lse; a.iso = (iso_len > 0 && iso_len < T_) ? iso_len : 0;
        return launch_fwd<float, false>(a, st);
    }
    AP<bf16_t> a = make_ap<bf16_t>(B, H, T_, q, k, v, ld, nullptr, 0, nullptr, nullptr, klen, 0, scale);
    a.o = (bf16_t*)o; a.ldo = ldo; a.lse = lse; a.iso = (iso_len > 0 && iso_len < T_) ? iso_len : 0;
    a.o_lo = (bf16_t*)o_lo;
    if (!attn_v1() && al8(o) && ldo % 4 == 0) return cvft_attn32_fwd(a, 0, st);
    if (o_lo) CVFT_HIP_CHECK_RET(hipMemsetAsync(o_lo, 0, (size_t)B * T_ * ldo * sizeof(bf16_t), st), "cvft_attn_bias_fwd");    // (the generic kernels write no residual)
    return launch_fwd<bf16_t, false>(a, st);
}

extern "C" int cvft_attn_bias_bwd(int dtype, int B, int H, int T_, const void* q, const void* k, const void* v, int ld,
                                  const int32_t* klen, float scale, int iso_len, const void* o, const void* d_o, int ldo,
                                  const float* lse, const void* o_lo, float* delta, void* dq, void* dk, void* dv, int ldg, void* stream) {
    if (check_common("cvft_attn_bias_bwd", dtype, B, H, T_, ld, ldo, q, k, v)) return -1;
    CVFT_CHECK_ARG(!o_lo || (dtype == CVFT_BF16 && (((uintptr_t)o_lo) & 15) == 0), "cvft_attn_bias_bwd: o_lo is a bf16 buffer (16-byte aligned) or NULL");
    CVFT_CHECK_ARG(d_o && lse && delta && dq && dk && dv && ldg >= H * 64, "cvft_attn_bias_bwd: bad args");
    // o == NULL: delta holds rowsum(dO . O) on entry (formed by the producer of dO); bf16 fused kernels only
    CVFT_CHECK_ARG(o || (dtype == CVFT_BF16 && !o_lo && !attn_v1() && al8(dq, dk, dv, d_o) && ldg % 4 == 0),
                   "cvft_attn_bias_bwd: o == NULL (delta given) needs bf16, no o_lo, 8-byte aligned gradients");
    CVFT_CHECK_ARG((((uintptr_t)d_o) & 15) == 0, "cvft_attn_bias_bwd: dO must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == CVFT_F32) {
        AP<float> a = make_ap<float>(B, H, T_, q, k, v, ld, nullptr, 0, nullptr, nullptr, klen, 0, scale);
        a.ldo = ldo; a.lse = (float*)lse; a.d_o = (const float*)d_o; a.delta = delta;
        a.dq = (float*)dq; a.dk = (float*)dk; a.dv = (float*)dv; a.ldg = ldg; a.iso = (iso_len > 0 && iso_len < T_) ? iso_len : 0;
        return launch_bwd<float, false>(a, delta, (const float*)o, st);
    }
    AP<bf16_t> a = make_ap<bf16_t>(B, H, T_, q, k, v, ld, nullptr, 0, nullptr, nullptr, klen, 0, scale);
    a.ldo = ldo; a.lse = (float*)lse; a.d_o = (const bf16_t*)d_o; a.delta = delta;
    a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv; a.ldg = ldg; a.iso = (iso_len > 0 && iso_len < T_) ? iso_len : 0;
    if (!o || (!attn_v1() && al8(dq, dk, dv, o) && ldg % 4 == 0)) {
        a.o = (bf16_t*)o;
        a.o_lo = (bf16_t*)o_lo;
        return cvft_attn32_bwd(a, 0, st);
    }
    return launch_bwd<bf16_t, false>(a, delta, (const bf16_t*)o, st);
}

extern "C" int cvft_attn_relpos_fwd(int dtype, int B, int H, int L, const void* q, const void* k, const void* v, int ld,
                                    const void* pp, int ldp, const float* bias_u, const float* bias_v,
                                    const int32_t* len, int causal, float scale, void* o, int ldo, float* lse, void* o_lo,
                                    float drop_p, const int64_t* drop_seed, unsigned drop_site, void* stream) {
    if (check_common("cvft_attn_relpos_fwd", dtype, B, H, L, ld, ldo, q, k, v)) return -1;
    CVFT_CHECK_ARG(!o_lo || (dtype == CVFT_BF16 && (((uintptr_t)o_lo) & 7) == 0), "cvft_attn_relpos_fwd: o_lo is a bf16 buffer (8-byte aligned) or NULL");
    CVFT_CHECK_ARG(cvft_drop_rate_ok(drop_p) && (drop_p == 0.f || drop_seed), "cvft_attn_relpos_fwd: bad dropout args (p == 0 or 2^-16 <= p <= 1 - 2^-16, seed)");
    int vec = dtype == CVFT_BF16 ? 8 : 4;
    CVFT_CHECK_ARG(pp && bias_u && bias_v && o && lse && ldp >= H * 64 && ldp % vec == 0 && (((uintptr_t)pp) & 15) == 0,
                   "cvft_attn_relpos_fwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == CVFT_F32) {
        AP<float> a = make_ap<float>(B, H, L, q, k, v, ld, pp, ldp, bias_u, bias_v, len, causal, scale);
        a.o = (float*)o; a.ldo = ldo; a.lse = lse;
        a.drop_p = drop_p; a.seed = (const long long*)drop_seed; a.site = drop_site;
        return launch_fwd<float, true>(a, st);
    }
    AP<bf16_t> a = make_ap<bf16_t>(B, H, L, q, k, v, ld, pp, ldp, bias_u, bias_v, len, causal, scale);
    a.o = (bf16_t*)o; a.ldo = ldo; a.lse = lse;
    a.drop_p = drop_p; a.seed = (const long long*)drop_seed; a.site = drop_site;
    a.o_lo = (bf16_t*)o_lo;
    if (!attn_v1() && al8(o) && ldo % 4 == 0) return cvft_attn32_fwd(a, 1, st);
    if (o_lo) CVFT_HIP_CHECK_RET(hipMemsetAsync(o_lo, 0, (size_t)B * L * ldo * sizeof(bf16_t), st), "cvft_attn_relpos_fwd");
    return launch_fwd<bf16_t, true>(a, st);
}

extern "C" int cvft_attn_relpos_bwd(int dtype, int B, int H, int L, const void* q, const void* k, const void* v, int ld,
                                    const void* pp, int ldp, const float* bias_u, const float* bias_v,
                                    const int32_t* len, int causal, float scale, const void* o, const void* d_o, int ldo,
                                    const float* lse, const void* o_lo, float* delta, void* dq, void* dk, void* dv, int ldg, float* dp,
                                    float drop_p, const int64_t* drop_seed, unsigned drop_site, void* stream) {
    if (check_common("cvft_attn_relpos_bwd", dtype, B, H, L, ld, ldo, q, k, v)) return -1;
    CVFT_CHECK_ARG(!o_lo || (dtype == CVFT_BF16 && (((uintptr_t)o_lo) & 15) == 0), "cvft_attn_relpos_bwd: o_lo is a bf16 buffer (16-byte aligned) or NULL");
    CVFT_CHECK_ARG(cvft_drop_rate_ok(drop_p) && (drop_p == 0.f || drop_seed), "cvft_attn_relpos_bwd: bad dropout args (p == 0 or 2^-16 <= p <= 1 - 2^-16, seed)");
    int vec = dtype == CVFT_BF16 ? 8 : 4;
    CVFT_CHECK_ARG(pp && bias_u && bias_v && ldp >= H * 64 && ldp % vec == 0 && (((uintptr_t)pp) & 15) == 0,
                   "cvft_attn_relpos_bwd: bad p");
    CVFT_CHECK_ARG(o && d_o && lse && delta && dq && dk && dv && ldg >= H * 64, "cvft_attn_relpos_bwd: bad args");
    CVFT_CHECK_ARG((((uintptr_t)d_o) & 15) == 0, "cvft_attn_relpos_bwd: dO must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == CVFT_F32) {
        AP<float> a = make_ap<float>(B, H, L, q, k, v, ld, pp, ldp, bias_u, bias_v, len, causal, scale);
        a.ldo = ldo; a.lse = (float*)lse; a.d_o = (const float*)d_o; a.delta = delta;
        a.dq = (float*)dq; a.dk = (float*)dk; a.dv = (float*)dv; a.ldg = ldg;
        a.drop_p = drop_p; a.seed = (const long long*)drop_seed; a.site = drop_site;
        a.dpos = dp; a.lddpos = H * 64;
        return launch_bwd<float, true>(a, delta, (const float*)o, st);
    }
    AP<bf16_t> a = make_ap<bf16_t>(B, H, L, q, k, v, ld, pp, ldp, bias_u, bias_v, len, causal, scale);
    a.ldo = ldo; a.lse = (float*)lse; a.d_o = (const bf16_t*)d_o; a.delta = delta;
    a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv; a.ldg = ldg;
    a.drop_p = drop_p; a.seed = (const long long*)drop_seed; a.site = drop_site;
    a.dpos = dp; a.lddpos = H * 64;
    if (!dp && !attn_v1() && al8(dq, dk, dv, o) && ldg % 4 == 0) {      // (dP is served by the generic kernels)
        a.o = (bf16_t*)o;
        a.o_lo = (bf16_t*)o_lo;
        return cvft_attn32_bwd(a, 1, st);
    }
    return launch_bwd<bf16_t, true>(a, delta, (const bf16_t*)o, st);
}
