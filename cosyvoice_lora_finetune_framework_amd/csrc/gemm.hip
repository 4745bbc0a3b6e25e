// gemm.hip -- tap-GEMM with fused rank-r LoRA side path and epilogue; LoRA-gradient
// "TN" accumulation.  MFMA 16x16 tiles (bf16 16x16x32 / exact-fp32 16x16x4), LDS-staged
// k-contiguous operand tiles, register-prefetched next tile (issue-early / write-late).
//
// Replaces (reference): lora.py:64-76, nn.Linear/Conv1d/ConvTranspose1d calls of
// modules.py:60-120 & matcha/models/components/decoder.py:35-158, and their dgrad.
#include "common.cuh"

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
};

template <typename T>
__device__ __forceinline__ uint4 load_chunk(const T* base, int kk, int klim, bool rowok, bool vec) {
    constexpr int VEC = 16 / sizeof(T);
    uint4 r = make_uint4(0, 0, 0, 0);
    if (!rowok || kk >= klim) return r;
    if (vec) return *reinterpret_cast<const uint4*>(base + kk);
    T tmp[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) tmp[e] = (kk + e < klim) ? base[kk + e] : from_f32<T>(0.f);
    return *reinterpret_cast<uint4*>(tmp);
}

template <typename T, int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(WM * WN * 64) gemm_kernel(GP<T> p) {
    constexpr int NT = WM * WN * 64;
    constexpr int VEC = 16 / sizeof(T);
    constexpr int BK = (sizeof(T) == 2) ? 64 : 16;
    constexpr int CPR = BK / VEC;
    constexpr int LD = BK + VEC;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 16, NI = TN / 16;
    constexpr int A_IT = (BM * CPR + NT - 1) / NT;
    constexpr int W_IT = (BN * CPR + NT - 1) / NT;
    typedef Mma<T> MM;

    __shared__ __attribute__((aligned(16))) T As[BM * LD];
    __shared__ __attribute__((aligned(16))) T Ws[BN * LD];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int m0 = (blockIdx.x / tiles_n) * BM;
    const int n0 = (blockIdx.x % tiles_n) * BN;

    // per-thread staging slots: row decode is loop invariant
    int a_row[A_IT], a_cc[A_IT], a_b[A_IT], a_t[A_IT], a_len[A_IT];
    bool a_ok[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        int c = tid + i * NT;
        a_row[i] = c / CPR;
        a_cc[i] = c % CPR;
        int m = m0 + a_row[i];
        a_ok[i] = (c < BM * CPR) && (m < p.M);
        int b = a_ok[i] ? m / p.Tm : 0;
        a_b[i] = b;
        a_t[i] = a_ok[i] ? m - b * p.Tm : 0;
        a_len[i] = (p.in_len && a_ok[i]) ? p.in_len[b] : p.Tin;
    }
    int w_row[W_IT], w_cc[W_IT];
    bool w_ok[W_IT];
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
        int c = tid + i * NT;
        w_row[i] = c / CPR;
        w_cc[i] = c % CPR;
        w_ok[i] = (c < BN * CPR) && (n0 + w_row[i] < p.N);
    }

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk_main = (p.K + BK - 1) / BK;
    const int nk_lora = (p.R > 0) ? (p.R + BK - 1) / BK : 0;
    const int n_it = p.ntaps * nk_main + nk_lora;

    uint4 ra[A_IT], rw[W_IT];

    auto gload = [&](int it) {
        if (it < p.ntaps * nk_main) {
            const int seg = it / nk_main;
            const int k0 = (it - seg * nk_main) * BK;
            const int toff = p.tap_off[seg];
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                int ti = a_t[i] * p.in_stride + toff;
                bool ok = a_ok[i] && ti >= 0 && ti < p.Tin && ti < a_len[i];
                const T* base = p.A + (size_t)(a_b[i] * p.Tin + (ok ? ti : 0)) * p.lda;
                ra[i] = load_chunk<T>(base, k0 + a_cc[i] * VEC, p.K, ok, p.vecA);
            }
#pragma unroll
            for (int i = 0; i < W_IT; ++i) {
                const T* base = p.W + (size_t)(w_ok[i] ? n0 + w_row[i] : 0) * p.ldw + (size_t)seg * p.K;
                rw[i] = load_chunk<T>(base, k0 + w_cc[i] * VEC, p.K, w_ok[i], p.vecW);
            }
        } else {
            const int k0 = (it - p.ntaps * nk_main) * BK;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                int m = m0 + a_row[i];
                const T* base = p.U + (size_t)(a_ok[i] ? m : 0) * p.ldu;
                ra[i] = load_chunk<T>(base, k0 + a_cc[i] * VEC, p.R, a_ok[i], p.vecU);
            }
#pragma unroll
            for (int i = 0; i < W_IT; ++i) {
                const T* base = p.Bl + (size_t)(w_ok[i] ? n0 + w_row[i] : 0) * p.ldbl;
                rw[i] = load_chunk<T>(base, k0 + w_cc[i] * VEC, p.R, w_ok[i], p.vecB);
            }
        }
    };
    auto sstore = [&]() {
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            if (tid + i * NT < BM * CPR) *reinterpret_cast<uint4*>(&As[a_row[i] * LD + a_cc[i] * VEC]) = ra[i];
#pragma unroll
        for (int i = 0; i < W_IT; ++i)
            if (tid + i * NT < BN * CPR) *reinterpret_cast<uint4*>(&Ws[w_row[i] * LD + w_cc[i] * VEC]) = rw[i];
    };

    gload(0);
    sstore();
    __syncthreads();
    for (int it = 0; it < n_it; ++it) {
        if (it + 1 < n_it) gload(it + 1);
#pragma unroll
        for (int ks = 0; ks < BK; ks += MM::K) {
            typename MM::Frag a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = MM::load(&As[(wm * TM + i * 16 + (lane & 15)) * LD + ks], lane);
#pragma unroll
            for (int j = 0; j < NI; ++j) b[j] = MM::load(&Ws[(wn * TN + j * 16 + (lane & 15)) * LD + ks], lane);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) MM::mma(acc[i][j], a[i], b[j]);
        }
        __syncthreads();
        if (it + 1 < n_it) {
            sstore();
            __syncthreads();
        }
    }

    // ------------------------------------------------------------- epilogue
    const bool ident = (p.Tm == p.M) && p.out_stride == 1 && p.out_off == 0;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int m = m0 + wm * TM + i * 16 + (lane >> 4) * 4 + r;
            if (m >= p.M) continue;
            int b = 0, to = m;
            if (!ident) {
                b = m / p.Tm;
                to = (m - b * p.Tm) * p.out_stride + p.out_off;
                if (to >= p.Tout) continue;
            }
            size_t orow = (size_t)b * p.Tout + to;
            bool live = p.out_len ? (to < p.out_len[b]) : true;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                int n = n0 + wn * TN + j * 16 + (lane & 15);
                if (n >= p.N) continue;
                float v = acc[i][j][r] * p.alpha;
                if (p.bias) v += p.bias[n];
                if (p.preact) p.preact[orow * p.ldp + n] = from_f32<T>(v);
                v = act_apply(p.act, v);
                if (p.dact_src) v *= act_grad(p.dact, to_f32(p.dact_src[orow * p.ldd + n]));
                if (p.residual) v += to_f32(p.residual[orow * p.ldr + n]);
                if (!live) v = 0.f;
                p.C[orow * p.ldc + n] = from_f32<T>(v);
            }
        }
    }
}

template <typename T>
static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename T>
static int gemm_launch(const cvft_gemm_args* a, hipStream_t st) {
    constexpr int VEC = 16 / sizeof(T);
    GP<T> p;
    p.M = a->M; p.N = a->N; p.K = a->K; p.Tm = a->Tm; p.Tin = a->Tin; p.Tout = a->Tout;
    p.in_stride = a->in_stride; p.out_stride = a->out_stride; p.out_off = a->out_off; p.ntaps = a->ntaps;
    for (int i = 0; i < 4; ++i) p.tap_off[i] = a->tap_off[i];
    p.in_len = a->in_len; p.out_len = a->out_len;
    p.A = (const T*)a->A; p.lda = a->lda; p.W = (const T*)a->W; p.ldw = a->ldw;
    p.U = (const T*)a->U; p.ldu = a->ldu; p.R = a->U ? a->R : 0; p.Bl = (const T*)a->Bl; p.ldbl = a->ldbl;
    p.bias = a->bias; p.alpha = a->alpha; p.act = a->act;
    p.preact = (T*)a->preact; p.ldp = a->ldp; p.dact_src = (const T*)a->dact_src; p.ldd = a->ldd; p.dact = a->dact;
    p.residual = (const T*)a->residual; p.ldr = a->ldr; p.C = (T*)a->C; p.ldc = a->ldc;
    p.vecA = (a->K % VEC == 0) && (a->lda % VEC == 0) && al16<T>(a->A);
    p.vecW = (a->K % VEC == 0) && (a->ldw % VEC == 0) && al16<T>(a->W);
    p.vecU = p.R > 0 && (p.R % VEC == 0) && (a->ldu % VEC == 0) && al16<T>(a->U);
    p.vecB = p.R > 0 && (p.R % VEC == 0) && (a->ldbl % VEC == 0) && al16<T>(a->Bl);

    auto tiles = [&](int bm, int bn) { return (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
    if (p.N <= 32) {
        hipLaunchKernelGGL((gemm_kernel<T, 32, 32, 2, 1>), dim3((unsigned)tiles(32, 32)), dim3(128), 0, st, p);
    } else if (tiles(128, 128) >= 512) {
        hipLaunchKernelGGL((gemm_kernel<T, 128, 128, 2, 2>), dim3((unsigned)tiles(128, 128)), dim3(256), 0, st, p);
    } else {
        hipLaunchKernelGGL((gemm_kernel<T, 64, 64, 2, 2>), dim3((unsigned)tiles(64, 64)), dim3(256), 0, st, p);
    }
    CVFT_LAUNCH_CHECK("cvft_gemm");
    return 0;
}

extern "C" int cvft_gemm(const cvft_gemm_args* a, void* stream) {
    CVFT_CHECK_ARG(a != nullptr, "cvft_gemm: null args");
    CVFT_CHECK_ARG(a->dtype == CVFT_F32 || a->dtype == CVFT_BF16, "cvft_gemm: bad dtype %d", a->dtype);
    CVFT_CHECK_ARG(a->M >= 0 && a->N > 0 && a->K > 0, "cvft_gemm: bad M/N/K %d %d %d", a->M, a->N, a->K);
    if (a->M == 0) return 0;
    CVFT_CHECK_ARG(a->ntaps >= 1 && a->ntaps <= 4, "cvft_gemm: ntaps %d", a->ntaps);
    CVFT_CHECK_ARG(a->Tm > 0 && a->M % a->Tm == 0, "cvft_gemm: M %d not a multiple of Tm %d", a->M, a->Tm);
    CVFT_CHECK_ARG(a->Tin > 0 && a->Tout > 0 && a->in_stride >= 1 && a->out_stride >= 1 && a->out_off >= 0,
                   "cvft_gemm: bad row geometry");
    CVFT_CHECK_ARG(a->A && a->W && a->C, "cvft_gemm: null operand");
    CVFT_CHECK_ARG(a->lda >= a->K && a->ldw >= a->ntaps * a->K && a->ldc >= a->N, "cvft_gemm: bad leading dims");
    if (a->U) {
        CVFT_CHECK_ARG(a->Bl && a->R > 0 && a->ldu >= a->R && a->ldbl >= a->R, "cvft_gemm: bad LoRA operands");
        CVFT_CHECK_ARG(a->ntaps == 1 && a->Tm == a->Tin && a->in_stride == 1 && a->tap_off[0] == 0,
                       "cvft_gemm: LoRA side path needs identity row geometry");
    }
    CVFT_CHECK_ARG(!a->preact || a->ldp >= a->N, "cvft_gemm: bad ldp");
    CVFT_CHECK_ARG(!a->dact_src || a->ldd >= a->N, "cvft_gemm: bad ldd");
    CVFT_CHECK_ARG(!a->residual || a->ldr >= a->N, "cvft_gemm: bad ldr");
    CVFT_CHECK_ARG((long)a->M * 1 < (1L << 31) && (long)(a->M / a->Tm) * a->Tin < (1L << 31), "cvft_gemm: too many rows");
    hipStream_t st = (hipStream_t)stream;
    return a->dtype == CVFT_F32 ? gemm_launch<float>(a, st) : gemm_launch<bf16_t>(a, st);
}

// ------------------------------------------------------------------------------
// G[p,q] += sum_m P[m,p] * Q[m,q]     (LoRA dA / dB; contraction over rows)
// 64x64 output tile per block, split over m; operands transposed while staging so the
// MFMA sees k(=m)-contiguous rows; fp32 atomics into G.
// ------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) tn_accum_kernel(int M, int P, int Q, const T* __restrict__ Pm, int ldp,
                                                        const T* __restrict__ Qm, int ldq, float* __restrict__ G,
                                                        int ldg, int m_per_block) {
    constexpr int BK = (sizeof(T) == 2) ? 64 : 16;
    constexpr int LD = BK + 16 / sizeof(T);
    typedef Mma<T> MM;
    __shared__ __attribute__((aligned(16))) T Pt[64 * LD];
    __shared__ __attribute__((aligned(16))) T Qt[64 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int tiles_q = (Q + 63) / 64;
    const int p0 = (blockIdx.x / tiles_q) * 64, q0 = (blockIdx.x % tiles_q) * 64;
    const int mb = blockIdx.y * m_per_block;
    const int me = min(M, mb + m_per_block);
    f32x4 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int mk = mb; mk < me; mk += BK) {
        // stage transposed: element (mm, c) -> Pt[c][mm]; consecutive threads walk c (coalesced rows)
        for (int e = tid; e < BK * 64; e += 256) {
            int mm = e >> 6, c = e & 63;
            int m = mk + mm;
            T pv = from_f32<T>(0.f), qv = from_f32<T>(0.f);
            if (m < me) {
                if (p0 + c < P) pv = Pm[(size_t)m * ldp + p0 + c];
                if (q0 + c < Q) qv = Qm[(size_t)m * ldq + q0 + c];
            }
            Pt[c * LD + mm] = pv;
            Qt[c * LD + mm] = qv;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < BK; ks += MM::K) {
            typename MM::Frag a[2], b[2];
            for (int i = 0; i < 2; ++i) a[i] = MM::load(&Pt[(wm * 32 + i * 16 + (lane & 15)) * LD + ks], lane);
            for (int j = 0; j < 2; ++j) b[j] = MM::load(&Qt[(wn * 32 + j * 16 + (lane & 15)) * LD + ks], lane);
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j) MM::mma(acc[i][j], a[i], b[j]);
        }
        __syncthreads();
    }
    for (int i = 0; i < 2; ++i)
        for (int r = 0; r < 4; ++r) {
            int pp = p0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r;
            if (pp >= P) continue;
            for (int j = 0; j < 2; ++j) {
                int qq = q0 + wn * 32 + j * 16 + (lane & 15);
                if (qq < Q) atomicAdd(&G[(size_t)pp * ldg + qq], acc[i][j][r]);
            }
        }
}

extern "C" int cvft_tn_accum(int dtype, int M, int P, int Q, const void* Pm, int ldp, const void* Qm, int ldq,
                             float* G, int ldg, void* stream) {
    CVFT_CHECK_ARG(dtype == CVFT_F32 || dtype == CVFT_BF16, "cvft_tn_accum: bad dtype");
    CVFT_CHECK_ARG(M >= 0 && P > 0 && Q > 0 && ldp >= P && ldq >= Q && ldg >= Q, "cvft_tn_accum: bad dims");
    CVFT_CHECK_ARG(Pm && Qm && G, "cvft_tn_accum: null operand");
    if (M == 0) return 0;
    int tiles = ((P + 63) / 64) * ((Q + 63) / 64);
    int splits = (1024 + tiles - 1) / tiles;
    int bk = dtype == CVFT_BF16 ? 64 : 16;
    int mpb = (M + splits - 1) / splits;
    mpb = ((mpb + bk - 1) / bk) * bk;
    if (mpb < 4 * bk) mpb = 4 * bk;
    splits = (M + mpb - 1) / mpb;
    dim3 grid(tiles, splits);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == CVFT_F32)
        hipLaunchKernelGGL((tn_accum_kernel<float>), grid, dim3(256), 0, st, M, P, Q, (const float*)Pm, ldp,
                           (const float*)Qm, ldq, G, ldg, mpb);
    else
        hipLaunchKernelGGL((tn_accum_kernel<bf16_t>), grid, dim3(256), 0, st, M, P, Q, (const bf16_t*)Pm, ldp,
                           (const bf16_t*)Qm, ldq, G, ldg, mpb);
    CVFT_LAUNCH_CHECK("cvft_tn_accum");
    return 0;
}

// ------------------------------------------------------------------------------
// LoRA adapter gradients, rank side r <= 64, VALU (HBM-bound: the wide operand is read once):
//   out[j, c] (+)= sum_m Rk[m, j] * Wd[m, c]      (transpose_out = 0 :  dA[r,K] = V^T X)
//   out[c, j] (+)= sum_m Rk[m, j] * Wd[m, c]      (transpose_out = 1 :  dB[N,r] = dY^T U)
// Block = 64 columns x 4 rank slices (one wavefront each); the [rows x r] rank chunk is staged in
// LDS as fp32 and read as broadcast ds_read; fp32 atomics combine the M-splits.
// ------------------------------------------------------------------------------
template <typename T, int RS>   // RS = rank values per thread (r = 4*RS)
__global__ void __launch_bounds__(256) lora_rank_accum_kernel(int M, int Cn, int r, const T* __restrict__ Wd, int ldw,
                                                               const T* __restrict__ Rk, int ldr, float* __restrict__ out,
                                                               int ldo, int transpose_out, int rows_per_block) {
    extern __shared__ float rk_s[];            // [CH][r]
    constexpr int CH = 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int mb = blockIdx.y * rows_per_block;
    const int me = min(M, mb + rows_per_block);
    float acc[RS];
#pragma unroll
    for (int j = 0; j < RS; ++j) acc[j] = 0.f;
    for (int m0 = mb; m0 < me; m0 += CH) {
        const int nrow = min(CH, me - m0);
        __syncthreads();
        for (int e = threadIdx.x; e < nrow * r; e += 256) rk_s[e] = to_f32(Rk[(size_t)(m0 + e / r) * ldr + (e % r)]);
        __syncthreads();
        if (c < Cn) {
            for (int mm = 0; mm < nrow; ++mm) {
                const float wv = to_f32(Wd[(size_t)(m0 + mm) * ldw + c]);
                const float* rp = rk_s + mm * r + w * RS;
#pragma unroll
                for (int j = 0; j < RS; ++j) acc[j] += wv * rp[j];
            }
        }
    }
    if (c < Cn) {
#pragma unroll
        for (int j = 0; j < RS; ++j) {
            const int jj = w * RS + j;
            float* dst = transpose_out ? &out[(size_t)c * ldo + jj] : &out[(size_t)jj * ldo + c];
            atomicAdd(dst, acc[j]);
        }
    }
}

template <typename T>
static int rank_accum_launch(int M, int Cn, int r, const void* Wd, int ldw, const void* Rk, int ldr, float* out, int ldo,
                             int transpose_out, hipStream_t st) {
    int colblocks = (Cn + 63) / 64;
    int splits = (768 + colblocks - 1) / colblocks;
    int rpb = (M + splits - 1) / splits;
    rpb = ((rpb + 63) / 64) * 64;
    if (rpb < 64) rpb = 64;
    splits = (M + rpb - 1) / rpb;
    dim3 grid(colblocks, splits);
    size_t sm = (size_t)64 * r * sizeof(float);
#define RA_LAUNCH(RS) hipLaunchKernelGGL((lora_rank_accum_kernel<T, RS>), grid, dim3(256), sm, st, M, Cn, r, (const T*)Wd, ldw, \
                                         (const T*)Rk, ldr, out, ldo, transpose_out, rpb)
    switch (r / 4) {
        case 1: RA_LAUNCH(1); break;
        case 2: RA_LAUNCH(2); break;
        case 4: RA_LAUNCH(4); break;
        case 8: RA_LAUNCH(8); break;
        case 16: RA_LAUNCH(16); break;
        default: return 1;
    }
#undef RA_LAUNCH
    return 0;
}

extern "C" int cvft_lora_rank_accum(int dtype, int M, int Cn, int r, const void* Wd, int ldw, const void* Rk, int ldr,
                                    float* out, int ldo, int transpose_out, void* stream) {
    CVFT_CHECK_ARG(dtype == CVFT_F32 || dtype == CVFT_BF16, "cvft_lora_rank_accum: bad dtype");
    CVFT_CHECK_ARG(M >= 0 && Cn > 0 && r > 0 && ldw >= Cn && ldr >= r && Wd && Rk && out, "cvft_lora_rank_accum: bad args");
    CVFT_CHECK_ARG(ldo >= (transpose_out ? r : Cn), "cvft_lora_rank_accum: bad ldo");
    if (M == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    int rc = 1;
    if (r % 4 == 0 && r <= 64) rc = dtype == CVFT_F32 ? rank_accum_launch<float>(M, Cn, r, Wd, ldw, Rk, ldr, out, ldo, transpose_out, st)
                                                      : rank_accum_launch<bf16_t>(M, Cn, r, Wd, ldw, Rk, ldr, out, ldo, transpose_out, st);
    if (rc == 1) {   // odd ranks: generic MFMA path  out = Rk^T Wd  or  Wd^T Rk
        return transpose_out ? cvft_tn_accum(dtype, M, Cn, r, Wd, ldw, Rk, ldr, out, ldo, stream)
                             : cvft_tn_accum(dtype, M, r, Cn, Rk, ldr, Wd, ldw, out, ldo, stream);
    }
    CVFT_LAUNCH_CHECK("cvft_lora_rank_accum");
    return 0;
}

// ------------------------------------------------------------------------------
// LoRA shadows: for every [rows x cols] fp32 master in the flat parameter buffer write a compute-dtype
// (bf16) copy at the same offset of `flat_c` and a TRANSPOSED copy ([cols x rows]) at the same offset
// of `flat_t` -- one launch per optimiser step for all adapters (32x32 LDS tiles).
//   tiles[t] = {param offset, rows, cols, tile_row * 65536 + tile_col}
// ------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) lora_shadow_kernel(const int4* __restrict__ tiles, const float* __restrict__ flat_p,
                                                           bf16_t* __restrict__ flat_c, bf16_t* __restrict__ flat_t) {
    __shared__ float tl[32][33];
    const int4 d = tiles[blockIdx.x];
    const int off = d.x, rows = d.y, cols = d.z, tr = (d.w >> 16) * 32, tc = (d.w & 0xffff) * 32;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;     // 32 x 8
    for (int i = ly; i < 32; i += 8) {
        int rr = tr + i, cc = tc + lx;
        float v = 0.f;
        if (rr < rows && cc < cols) {
            v = flat_p[(size_t)off + (size_t)rr * cols + cc];
            flat_c[(size_t)off + (size_t)rr * cols + cc] = (bf16_t)v;
        }
        tl[i][lx] = v;
    }
    __syncthreads();
    for (int i = ly; i < 32; i += 8) {
        int cc = tc + i, rr = tr + lx;
        if (rr < rows && cc < cols) flat_t[(size_t)off + (size_t)cc * rows + rr] = (bf16_t)tl[lx][i];
    }
}
extern "C" int cvft_lora_shadow(int ntiles, const void* tiles, const float* flat_p, void* flat_c, void* flat_t, void* stream) {
    CVFT_CHECK_ARG(ntiles >= 0 && tiles && flat_p && flat_c && flat_t, "cvft_lora_shadow: bad args");
    if (ntiles == 0) return 0;
    hipLaunchKernelGGL(lora_shadow_kernel, dim3(ntiles), dim3(256), 0, (hipStream_t)stream, (const int4*)tiles, flat_p,
                       (bf16_t*)flat_c, (bf16_t*)flat_t);
    CVFT_LAUNCH_CHECK("cvft_lora_shadow");
    return 0;
}
