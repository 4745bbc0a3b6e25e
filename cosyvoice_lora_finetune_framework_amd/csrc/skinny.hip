// skinny.hip -- bf16 rank-side GEMM  C[M, R] = alpha * X[M, K] . A[R, K]^T  with R in {16, 32, 48, 64}  (the LoRA
// "U = s x A^T" / "V = s dY B" products).  HBM-bound (X is read once, 2*M*K bytes; A is a few KB and L2-resident):
// a 64x64-tile GEMM kernel gives it M/64 blocks (63 for the estimator) and walks K serially -- pure latency.
// Here one workgroup owns 32 rows; its 8 wavefronts split K, every wave keeps several k-steps of fragment-shaped
// loads in flight (16 rows x 64 B per instruction, straight into the MFMA operand registers, no LDS staging), and
// the eight partial tiles are summed through LDS.  M/32 blocks x 8 waves: 1000+ waves in flight.
//
// Replaces (reference): the x @ lora_A.T product of lora.py:71-73 (and its transpose in backward).
#include <stdlib.h>
#include "gemm_common.h"

template <int RB, int MT, int KS>      // RB = R / 16 column tiles, MT = 16-row tiles per block, KS = k-steps in flight per wave
__global__ void __launch_bounds__(512) skinny_kernel(int M, int K, const bf16_t* __restrict__ X, int ldx,
                                                     const bf16_t* __restrict__ A, int lda, float alpha,
                                                     bf16_t* __restrict__ C, int ldc) {
    // 8 wavefronts split K; every wave covers all MT row tiles so that each A fragment it loads is used MT times
    // (at one row tile per block the re-reads of A from L2 exceeded the X stream itself)
    __shared__ __attribute__((aligned(16))) float red[4][MT][RB][16][17];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l15 = lane & 15, kg = lane >> 4;
    const int m0 = blockIdx.x * 16 * MT;
    const int kper = ((K / 32 + 7) / 8) * 32;                    // this wave's K slice [kb, ke)
    const int kb = w * kper, ke = min(K, kb + kper);
    f32x4 acc[MT][RB];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int j = 0; j < RB; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16_t* xp[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) xp[t] = X + (size_t)min(m0 + t * 16 + l15, M - 1) * ldx + kg * 8;   // clamped: rows >= M never stored
    const bf16_t* ap = A + (size_t)l15 * lda + kg * 8;
    for (int k0 = kb; k0 < ke; k0 += 32 * KS) {
        uint4 xv[KS][MT], av[KS][RB];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k = min(k0 + 32 * s, K - 32);              // clamped address, masked below (no branch around a load)
#pragma unroll
            for (int t = 0; t < MT; ++t) xv[s][t] = *reinterpret_cast<const uint4*>(xp[t] + k);
#pragma unroll
            for (int j = 0; j < RB; ++j) av[s][j] = *reinterpret_cast<const uint4*>(ap + (size_t)j * 16 * lda + k);
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const unsigned keep = (k0 + 32 * s < ke) ? 0xffffffffu : 0u;
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                uint4 v = av[s][j];
                v.x &= keep; v.y &= keep; v.z &= keep; v.w &= keep;
                const bf16x8 b = *reinterpret_cast<bf16x8*>(&v);
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&xv[s][t]), b, acc[t][j], 0, 0, 0);
            }
        }
    }
    // 8 partial tiles -> 4 (waves 4..7 hand theirs to waves 0..3) -> final sum
    if (w >= 4) {
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int j = 0; j < RB; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[w - 4][t][j][kg * 4 + r][l15] = acc[t][j][r];
    }
    __syncthreads();
    if (w < 4) {
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int j = 0; j < RB; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[w][t][j][kg * 4 + r][l15] += acc[t][j][r];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 16 * MT * RB * 16; e += 512) {
        const int r = e / (RB * 16), c = e % (RB * 16);
        if (m0 + r >= M) continue;
        const int t = r >> 4, rr = r & 15, j = c >> 4, cc = c & 15;
        const float s = red[0][t][j][rr][cc] + red[1][t][j][rr][cc] + red[2][t][j][rr][cc] + red[3][t][j][rr][cc];
        C[(size_t)(m0 + r) * ldc + c] = (bf16_t)(alpha * s);
    }
}

// returns 1 when the launch is not a plain skinny product
int skinny_launch(const GP<bf16_t>& p, hipStream_t st) {
    const bool ident = p.ntaps == 1 && p.tap_off[0] == 0 && p.in_stride == 1 && p.Tin == p.Tm && !p.in_len && p.Tm == p.M &&
                       p.out_stride == 1 && p.out_off == 0 && !p.out_len;
    if (!ident || p.fuse || p.R > 0 || p.bias || p.act || p.preact || p.dact_src || p.residual) return 1;
    if (p.N > 64 || p.N % 16 != 0 || p.K % 32 != 0 || p.K < 32 || !p.vecA || !p.vecW) return 1;
    // rows per block: 32 (every A fragment used twice) unless that leaves half the chip without a block -- measured
    // (tools/bench_side.py, cold): M = 4000, K = 1536, R = 48: 12.3 us at 32 rows (125 blocks), 10.0 at 16 (250), 16.6 at 64;
    // from M = 5376 up 16 rows lose (20.2 -> 25.7 us at K = 3072: A is re-read from L2 once per block)
    static const int mt_forced = getenv("CVFT_SKINNY_MT") ? atoi(getenv("CVFT_SKINNY_MT")) : 0;
    const int mt_env = mt_forced ? mt_forced : (p.M <= 4096 ? 1 : 2);
    const int ksteps_per_wave = (p.K / 32 + 7) / 8;
#define SK_LAUNCH(RBv, MTv, KSv) hipLaunchKernelGGL((skinny_kernel<RBv, MTv, KSv>), dim3((p.M + 16 * MTv - 1) / (16 * MTv)), dim3(512), 0, st, p.M, p.K, p.A, p.lda, p.W, p.ldw, \
                                               p.alpha, p.C, p.ldc)
#define SK_KS(RBv, MTv) do { if (ksteps_per_wave >= 4) SK_LAUNCH(RBv, MTv, 4); else if (ksteps_per_wave >= 2) SK_LAUNCH(RBv, MTv, 2); else SK_LAUNCH(RBv, MTv, 1); } while (0)
#define SK_RB(RBv) do { if (mt_env == 1) SK_KS(RBv, 1); else if (mt_env == 4) SK_KS(RBv, 4); else SK_KS(RBv, 2); } while (0)
    switch (p.N / 16) {
        case 1: SK_RB(1); break;
        case 2: SK_RB(2); break;
        case 3: SK_RB(3); break;
        default: SK_RB(4); break;
    }
#undef SK_RB
#undef SK_KS
#undef SK_LAUNCH
    cvft_set_kernel_label("skinny_kernel<bf16,r%d>", p.N);
    CVFT_LAUNCH_CHECK("cvft_gemm");
    return 0;
}


// ------------------------------------------------------------------------------
// The same product on a DROPPED input (lora.py:70-73 in train mode):  C[:, 16t:16t+16] = alpha/(1-p) * drop_t(X) . A_t^T,
// one mask site per rank tile t (the stacked q|k|v adapters each have their own nn.Dropout).  The mask is applied to
// the X fragments in registers (bitwise AND; the 1/(1-p) scale rides in alpha), so drop(x) is never materialised.
// ------------------------------------------------------------------------------
// SHARED: every rank tile uses the mask of site 0 (ONE adapter of rank 16*RB, e.g. r = 64 of BASELINE configs[4]): the mask
// is derived once per fragment instead of once per rank tile, and drop(X) is written once (xd0).
// LN: X is the INPUT of a LayerNorm (encoder_layer.py:90-104, matcha transformer.py:255-316 pre-norm blocks) whose output
// feeds this adapter: the block owns whole rows (its 8 waves split K), so the row statistics are two LDS reductions away --
// the kernel normalises its fragments in registers (two-pass statistics like ln_fwd_kernel), writes Y = LN(X) for the main
// GEMM plus mean / rstd for the LayerNorm backward, and carries on with the masked product on Y.  One launch instead of
// LayerNorm + rank-side product, and LN(X) is not read back.  Needs the wave's whole K slice in one pass (K <= 256 * KS).
struct LnArgs { const float* gamma; const float* beta; float eps; bf16_t* Y; float* mean; float* rstd; };
template <int RB, int MT, int KS, bool SHARED = false, bool LN = false>
__global__ void __launch_bounds__(512) skinny_dropout_kernel(int M, int K, const bf16_t* __restrict__ X, int ldx,
                                                             const bf16_t* __restrict__ A, int lda, float alpha,
                                                             bf16_t* __restrict__ C, int ldc, float p,
                                                             const long long* __restrict__ seed, uint4 sites,
                                                             bf16_t* __restrict__ xd0, bf16_t* __restrict__ xd1,
                                                             bf16_t* __restrict__ xd2, LnArgs ln) {
    // xd0..2 (optional, one per rank tile / mask site): the dropped input drop_t(X) = keep_t * X / (1 - p) is also written
    // out, [M][K] each -- the backward pass needs it for dA_t = V_t^T drop_t(X) and would otherwise re-derive it in a pass
    // of its own (one launch per adapter per step)
    __shared__ __attribute__((aligned(16))) float red[4][MT][RB][16][17];
    __shared__ float lnred[2][8][MT][16];
    bf16_t* const xd[3] = {xd0, xd1, xd2};
    const float inv_keep = 1.f / (1.f - p);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l15 = lane & 15, kg = lane >> 4;
    const int m0 = blockIdx.x * 16 * MT;
    const int kper = ((K / 32 + 7) / 8) * 32;
    const int kb = w * kper, ke = min(K, kb + kper);
    const unsigned thr = cvft_drop_thr(p);
    const unsigned st[4] = {sites.x, sites.y, sites.z, sites.w};
    unsigned long long keys[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) keys[j] = cvft_drop_key(seed, st[j]);
    f32x4 acc[MT][RB];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int j = 0; j < RB; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int rows[MT];
    const bf16_t* xp[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        rows[t] = min(m0 + t * 16 + l15, M - 1);
        xp[t] = X + (size_t)rows[t] * ldx + kg * 8;
    }
    const bf16_t* ap = A + (size_t)l15 * lda + kg * 8;
    // LN: exactly one pass, taken by every wave (also one whose K slice is empty: it still meets the barriers)
    for (int k0 = kb, pass = 0; LN ? pass < 1 : k0 < ke; k0 += 32 * KS, ++pass) {
        uint4 xv[KS][MT], av[KS][RB];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k = min(k0 + 32 * s, K - 32);
#pragma unroll
            for (int t = 0; t < MT; ++t) xv[s][t] = *reinterpret_cast<const uint4*>(xp[t] + k);
#pragma unroll
            for (int j = 0; j < RB; ++j) av[s][j] = *reinterpret_cast<const uint4*>(ap + (size_t)j * 16 * lda + k);
        }
        if constexpr (LN) {
            float4 gv[KS][2], bv[KS][2];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int k = min(k0 + 32 * s, K - 32) + kg * 8;
                gv[s][0] = *reinterpret_cast<const float4*>(ln.gamma + k); gv[s][1] = *reinterpret_cast<const float4*>(ln.gamma + k + 4);
                bv[s][0] = *reinterpret_cast<const float4*>(ln.beta + k); bv[s][1] = *reinterpret_cast<const float4*>(ln.beta + k + 4);
            }
            float mu[MT], rs[MT];
            // pass 1: mean.  lane (kg, l15) holds 8 consecutive k of row l15 per k-step; the 4 kg lanes, then the 8 waves
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                float sm = 0.f;
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    if (k0 + 32 * s < ke) {
                        const bf16x8 v = *reinterpret_cast<const bf16x8*>(&xv[s][t]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) sm += (float)v[e];
                    }
                sm += __shfl_xor(sm, 16);
                sm += __shfl_xor(sm, 32);
                if (kg == 0) lnred[0][w][t][l15] = sm;
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                float sm = 0.f;
#pragma unroll
                for (int ww = 0; ww < 8; ++ww) sm += lnred[0][ww][t][l15];
                mu[t] = sm / (float)K;
            }
            // pass 2: centred variance
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                float sq = 0.f;
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    if (k0 + 32 * s < ke) {
                        const bf16x8 v = *reinterpret_cast<const bf16x8*>(&xv[s][t]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float d = (float)v[e] - mu[t]; sq += d * d; }
                    }
                sq += __shfl_xor(sq, 16);
                sq += __shfl_xor(sq, 32);
                if (kg == 0) lnred[1][w][t][l15] = sq;
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                float sq = 0.f;
#pragma unroll
                for (int ww = 0; ww < 8; ++ww) sq += lnred[1][ww][t][l15];
                rs[t] = 1.0f / sqrtf(sq / (float)K + ln.eps);
                if (w == 0 && kg == 0 && m0 + t * 16 + l15 < M) { ln.mean[rows[t]] = mu[t]; ln.rstd[rows[t]] = rs[t]; }
            }
            // normalise in place (what the main GEMM reads is exactly what the masked product sees) and publish Y
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int k = min(k0 + 32 * s, K - 32) + kg * 8;
                const float g[8] = {gv[s][0].x, gv[s][0].y, gv[s][0].z, gv[s][0].w, gv[s][1].x, gv[s][1].y, gv[s][1].z, gv[s][1].w};
                const float b[8] = {bv[s][0].x, bv[s][0].y, bv[s][0].z, bv[s][0].w, bv[s][1].x, bv[s][1].y, bv[s][1].z, bv[s][1].w};
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const bf16x8 v = *reinterpret_cast<const bf16x8*>(&xv[s][t]);
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(((float)v[e] - mu[t]) * rs[t] * g[e] + b[e]);
                    xv[s][t] = *reinterpret_cast<const uint4*>(&o);
                    if (k0 + 32 * s < ke && m0 + t * 16 + l15 < M) *reinterpret_cast<bf16x8*>(ln.Y + (size_t)rows[t] * K + k) = o;
                }
            }
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bool live = k0 + 32 * s < ke;
            const int k = min(k0 + 32 * s, K - 32) + kg * 8;     // first of this lane's 8 consecutive k
            auto masked = [&](int t, unsigned long long key) __attribute__((always_inline)) {
                const unsigned long long g = ((unsigned long long)rows[t] * K + k) >> 2;
                bool k0_[4], k1_[4];
                cvft_keep4(key, g, thr, k0_);
                cvft_keep4(key, g + 1, thr, k1_);
                uint4 v = xv[s][t];
                v.x &= (live && k0_[0] ? 0x0000ffffu : 0u) | (live && k0_[1] ? 0xffff0000u : 0u);
                v.y &= (live && k0_[2] ? 0x0000ffffu : 0u) | (live && k0_[3] ? 0xffff0000u : 0u);
                v.z &= (live && k1_[0] ? 0x0000ffffu : 0u) | (live && k1_[1] ? 0xffff0000u : 0u);
                v.w &= (live && k1_[2] ? 0x0000ffffu : 0u) | (live && k1_[3] ? 0xffff0000u : 0u);
                return *reinterpret_cast<bf16x8*>(&v);
            };
            auto keep = [&](int t, int j, const bf16x8& vm) __attribute__((always_inline)) {
                if (xd[j] && live && m0 + t * 16 + l15 < M) {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)vm[e] * inv_keep);
                    *reinterpret_cast<bf16x8*>(xd[j] + (size_t)rows[t] * K + k) = o;
                }
            };
            if (SHARED) {
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const bf16x8 vm = masked(t, keys[0]);
                    keep(t, 0, vm);
#pragma unroll
                    for (int j = 0; j < RB; ++j)
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vm, *reinterpret_cast<bf16x8*>(&av[s][j]), acc[t][j], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < RB; ++j) {
                    const bf16x8 b = *reinterpret_cast<bf16x8*>(&av[s][j]);
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        const bf16x8 vm = masked(t, keys[j]);
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vm, b, acc[t][j], 0, 0, 0);
                        if (j < 3) keep(t, j < 3 ? j : 0, vm);
                    }
                }
            }
        }
    }
    if (w >= 4) {
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int j = 0; j < RB; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[w - 4][t][j][kg * 4 + r][l15] = acc[t][j][r];
    }
    __syncthreads();
    if (w < 4) {
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int j = 0; j < RB; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[w][t][j][kg * 4 + r][l15] += acc[t][j][r];
    }
    __syncthreads();
    const float sc = alpha / (1.f - p);
    for (int e = threadIdx.x; e < 16 * MT * RB * 16; e += 512) {
        const int r = e / (RB * 16), c = e % (RB * 16);
        if (m0 + r >= M) continue;
        const int t = r >> 4, rr = r & 15, j = c >> 4, cc = c & 15;
        const float s = red[0][t][j][rr][cc] + red[1][t][j][rr][cc] + red[2][t][j][rr][cc] + red[3][t][j][rr][cc];
        C[(size_t)(m0 + r) * ldc + c] = (bf16_t)(sc * s);
    }
}

// U[M, R] = alpha * sum_k drop_t(X)[m,k] A[16t + j][k]   (X contiguous rows of K: the mask index is m*K + k)
extern "C" int cvft_skinny_dropout(int M, int K, int R, const void* X, int ldx, const void* A, int lda, float alpha, void* C,
                                   int ldc, float p, const int64_t* seed, const unsigned* sites, void* const* xd, void* stream) {
    CVFT_CHECK_ARG(M > 0 && K >= 32 && K % 32 == 0 && (R == 16 || R == 32 || R == 48 || R == 64) && X && A && C && seed && sites &&
                   ldx == K && lda >= K && ldc >= R && p > 0.f && cvft_drop_rate_ok(p) && (((uintptr_t)X | (uintptr_t)A) & 15) == 0 && lda % 8 == 0,
                   "cvft_skinny_dropout: bad args (bf16, contiguous X rows, K %% 32 == 0, R in {16, 32, 48, 64})");
    const int nt = R / 16;
    bool shared = true;                 // one adapter: every rank tile under the same mask site
    for (int t = 1; t < nt; ++t) shared = shared && sites[t] == sites[0];
    CVFT_CHECK_ARG(shared || R == 48, "cvft_skinny_dropout: distinct mask sites per rank tile only for R = 48 (stacked q|k|v)");
    constexpr int MT = 2;
    dim3 grid((M + 16 * MT - 1) / (16 * MT));
    const int ksteps_per_wave = (K / 32 + 7) / 8;
    uint4 st = make_uint4(sites[0], nt > 1 ? sites[1] : 0u, nt > 2 ? sites[2] : 0u, nt > 3 ? sites[3] : 0u);
    const bool per_tile = !shared;
    const LnArgs noln = {nullptr, nullptr, 0.f, nullptr, nullptr, nullptr};
#define SKD_LAUNCH(RBv, KSv, SHv) hipLaunchKernelGGL((skinny_dropout_kernel<RBv, MT, KSv, SHv>), grid, dim3(512), 0, (hipStream_t)stream, M, K, \
                                                (const bf16_t*)X, ldx, (const bf16_t*)A, lda, alpha, (bf16_t*)C, ldc, p, (const long long*)seed, st, \
                                                (bf16_t*)(xd ? xd[0] : nullptr), (bf16_t*)(xd && per_tile ? xd[1] : nullptr), (bf16_t*)(xd && per_tile ? xd[2] : nullptr), noln)
    const bool k2 = ksteps_per_wave >= 2;
    if (R == 16) { if (k2) SKD_LAUNCH(1, 2, false); else SKD_LAUNCH(1, 1, false); }
    else if (per_tile) { if (k2) SKD_LAUNCH(3, 2, false); else SKD_LAUNCH(3, 1, false); }
    else if (R == 32) { if (k2) SKD_LAUNCH(2, 2, true); else SKD_LAUNCH(2, 1, true); }
    else if (R == 48) { if (k2) SKD_LAUNCH(3, 2, true); else SKD_LAUNCH(3, 1, true); }
    else { if (k2) SKD_LAUNCH(4, 2, true); else SKD_LAUNCH(4, 1, true); }
#undef SKD_LAUNCH
    CVFT_LAUNCH_CHECK("cvft_skinny_dropout");
    return 0;
}

// LayerNorm + the dropped rank-side product of its output in ONE launch (see skinny_dropout_kernel, LN):
//   Y = LN(X) (bf16), mean / rstd [M] (fp32, for cvft_layernorm_bwd), U = alpha / (1 - p) * drop_t(Y) A_t^T, xd[t] = drop_t(Y)
extern "C" int cvft_ln_skinny_dropout(int M, int K, int R, const void* X, const float* gamma, const float* beta, float eps, void* Y,
                                      float* mean, float* rstd, const void* A, int lda, float alpha, void* U, int ldu, float p,
                                      const int64_t* seed, const unsigned* sites, void* const* xd, void* stream) {
    CVFT_CHECK_ARG(M > 0 && K >= 32 && K % 32 == 0 && K <= 1024 && (R == 16 || R == 48) && X && gamma && beta && Y && mean && rstd && A && U &&
                   seed && sites && lda >= K && ldu >= R && p > 0.f && cvft_drop_rate_ok(p) && lda % 8 == 0 &&
                   (((uintptr_t)X | (uintptr_t)A | (uintptr_t)Y | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0,
                   "cvft_ln_skinny_dropout: bad args (bf16 contiguous rows, K %% 32 == 0, K <= 1024, R in {16, 48}, 16-byte aligned operands)");
    const int nt = R / 16;
    bool shared = true;
    for (int t = 1; t < nt; ++t) shared = shared && sites[t] == sites[0];
    CVFT_CHECK_ARG(nt == 1 || !shared, "cvft_ln_skinny_dropout: R = 48 is the stacked q|k|v form (three distinct mask sites)");
    const int ksteps_per_wave = (K / 32 + 7) / 8;              // 1 .. 4: the wave's whole K slice stays in registers
    uint4 st = make_uint4(sites[0], nt > 1 ? sites[1] : 0u, nt > 2 ? sites[2] : 0u, 0u);
    const LnArgs ln = {gamma, beta, eps, (bf16_t*)Y, mean, rstd};
    // 16 rows per block while 32 would leave half the chip without a block (the estimator's half batches), like skinny_launch
    static const int mt_forced = getenv("CVFT_SKINNY_MT") ? atoi(getenv("CVFT_SKINNY_MT")) : 0;
    const int mt = mt_forced ? mt_forced : (M <= 4096 ? 1 : 2);
#define LSK_LAUNCH(RBv, MTv, KSv) hipLaunchKernelGGL((skinny_dropout_kernel<RBv, MTv, KSv, false, true>), dim3((M + 16 * MTv - 1) / (16 * MTv)), dim3(512), 0, \
                                           (hipStream_t)stream, M, K,                                                                           \
                                           (const bf16_t*)X, K, (const bf16_t*)A, lda, alpha, (bf16_t*)U, ldu, p, (const long long*)seed, st,      \
                                           (bf16_t*)(xd ? xd[0] : nullptr), (bf16_t*)(xd && nt > 1 ? xd[1] : nullptr),                          \
                                           (bf16_t*)(xd && nt > 1 ? xd[2] : nullptr), ln)
#define LSK_KS(RBv, MTv) do { if (ksteps_per_wave <= 1) LSK_LAUNCH(RBv, MTv, 1); else if (ksteps_per_wave == 2) LSK_LAUNCH(RBv, MTv, 2); else LSK_LAUNCH(RBv, MTv, 4); } while (0)
#define LSK_MT(RBv) do { if (mt == 1) LSK_KS(RBv, 1); else LSK_KS(RBv, 2); } while (0)
    if (R == 16) LSK_MT(1); else LSK_MT(3);
#undef LSK_MT
#undef LSK_KS
#undef LSK_LAUNCH
    CVFT_LAUNCH_CHECK("cvft_ln_skinny_dropout");
    return 0;
}
