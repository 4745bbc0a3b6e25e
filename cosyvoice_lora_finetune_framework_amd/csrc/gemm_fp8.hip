// gemm_fp8.hip -- OCP fp8 (e4m3) building blocks for BASELINE configs[4] (frozen-W GEMMs in fp8).
// Round 1: the block-scaled matrix instruction's operand layout, pinned by an exact-data probe
// (v_mfma_scale_f32_16x16x128_f8f6f4 with unit e8m0 scales = plain e4m3 x e4m3 -> f32 at twice the bf16 rate).
#include "common.h"

typedef int v8i_t __attribute__((ext_vector_type(8)));

// One 16x16x128 product: C[i][j] = sum_k A[i][k] * B[j][k], A and B as [16][128] e4m3 bytes (k contiguous).
// Operand layout under test: lane l supplies row l & 15, bytes 32 * (l >> 4) .. + 31 of that row (8 dwords).
__global__ void mfma_fp8_probe_kernel(const unsigned char* __restrict__ A, const unsigned char* __restrict__ B, float* __restrict__ C) {
    const int lane = threadIdx.x;
    v8i_t a, b;
    const int* ap = reinterpret_cast<const int*>(A + (lane & 15) * 128 + (lane >> 4) * 32);
    const int* bp = reinterpret_cast<const int*>(B + (lane & 15) * 128 + (lane >> 4) * 32);
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = ap[i]; b[i] = bp[i]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0 /*A: e4m3*/, 0 /*B: e4m3*/, 0, 0x7f7f7f7f /*scale 2^0*/, 0, 0x7f7f7f7f);
#pragma unroll
    for (int r = 0; r < 4; ++r) C[((lane >> 4) * 4 + r) * 16 + (lane & 15)] = c[r];
}

extern "C" int cvft_debug_mfma_fp8_probe(const void* A, const void* B, float* C, void* stream) {
    CVFT_CHECK_ARG(A && B && C, "cvft_debug_mfma_fp8_probe: null operand");
    hipLaunchKernelGGL(mfma_fp8_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned char*)A, (const unsigned char*)B, C);
    CVFT_LAUNCH_CHECK("cvft_debug_mfma_fp8_probe");
    return 0;
}

// ------------------------------------------------------------------------------
// Row-wise e4m3 quantisation:  q[m][k] = e4m3(x[m][k] / s[m]),  s[m] = max_k |x[m][k]| / 448  (1 for an all-zero row).
// Rows of x = activation rows (per-token scale) or weight rows (per-output-channel scale).  One wavefront per row,
// 16-byte loads, the row is read twice (the second pass hits L2), fp8 written 8 bytes per lane and step.
// ------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) quant_fp8_rows_kernel(int M, int K, const bf16_t* __restrict__ x, int ldx,
                                                             unsigned char* __restrict__ q, int ldq, float* __restrict__ scale) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const bf16_t* xr = x + (size_t)row * ldx;
    float amax = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
        const uint4 v = *reinterpret_cast<const uint4*>(xr + k);
        const bf16_t* e = reinterpret_cast<const bf16_t*>(&v);
#pragma unroll
        for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf((float)e[i]));
    }
    amax = wave_max(amax);
    const float s = amax > 0.f ? amax * (1.f / 448.f) : 1.f;
    const float inv = 1.f / s;
    if (lane == 0) scale[row] = s;
    unsigned char* qr = q + (size_t)row * ldq;
    for (int k = lane * 8; k < K; k += 512) {
        const uint4 v = *reinterpret_cast<const uint4*>(xr + k);
        const bf16_t* e = reinterpret_cast<const bf16_t*>(&v);
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32((float)e[0] * inv, (float)e[1] * inv, lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32((float)e[2] * inv, (float)e[3] * inv, lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32((float)e[4] * inv, (float)e[5] * inv, hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32((float)e[6] * inv, (float)e[7] * inv, hi, true);
        *reinterpret_cast<uint2*>(qr + k) = make_uint2((unsigned)lo, (unsigned)hi);
    }
}

extern "C" int cvft_quant_fp8_rows(int M, int K, const void* x, int ldx, void* q, int ldq, float* scale, void* stream) {
    CVFT_CHECK_ARG(M > 0 && K > 0 && K % 8 == 0 && x && q && scale && ldx >= K && ldx % 8 == 0 && ldq >= K && ldq % 8 == 0 &&
                   (((uintptr_t)x & 15) == 0) && (((uintptr_t)q & 7) == 0),
                   "cvft_quant_fp8_rows: bf16 rows, K %% 8 == 0, 16-byte aligned x, 8-byte aligned q");
    hipLaunchKernelGGL(quant_fp8_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, M, K, (const bf16_t*)x, ldx,
                       (unsigned char*)q, ldq, scale);
    CVFT_LAUNCH_CHECK("cvft_quant_fp8_rows");
    return 0;
}

// ------------------------------------------------------------------------------
// fp8 GEMM:   C[M,N] = epilogue( alpha * ( sa[m] * sw[n] * (A8[M,K] . W8[N,K]^T)  +  U[M,R] . Bl[N,R]^T ) )
// A8 / W8: e4m3 bytes with per-row scales (cvft_quant_fp8_rows); U / Bl (the LoRA side path), bias, activation, residual
// and the output stay bf16 / fp32 exactly as in cvft_gemm.  Structure = the 128 x 128 LDS-DMA kernel of gemm_glds.hip
// (8 waves as 4 x 2, two 32 KB stages, XOR-swizzled 128-byte rows, register epilogue) with 128 k per tile instead of 64:
// the same bytes from L2 carry twice the FLOPs and the block-scaled MFMA retires them at twice the bf16 rate.
// ------------------------------------------------------------------------------
#include "gemm_common.h"
typedef __attribute__((address_space(3))) void lds_void8_t;
typedef const __attribute__((address_space(1))) void glb_void8_t;

__global__ void __launch_bounds__(512) gemm_fp8_kernel(GP<bf16_t> p, const unsigned char* __restrict__ A8, int lda8,
                                                       const float* __restrict__ sa, const unsigned char* __restrict__ W8,
                                                       int ldw8, const float* __restrict__ sw) {
    typedef bf16_t T;
    constexpr int BM = 128, BN = 128, WM = 4, WN = 2, NW = 8, BKB = 128;      // BKB: tile depth in BYTES = fp8 elements
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 16, NI = TN / 16;   // 32 x 64 wave tile
    constexpr int A_BYTES = BM * 128, BUF = (BM + BN) * 128;
    constexpr int A_INS = BM / 8 / NW, W_INS = BN / 8 / NW;                   // 2 + 2 DMA pieces per wave and tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int tiles_n = (p.N + BN - 1) / BN;
    int m0, n0;
    {
        int bid = blockIdx.x;
        const int nwg = gridDim.x, q = nwg >> 3, rm = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + loc;      // XCD-aware bijective remap, n fastest
        m0 = (bid / tiles_n) * BM;
        n0 = (bid % tiles_n) * BN;
    }
    const unsigned char* ga[A_INS];
    const unsigned char* gw[W_INS];
#pragma unroll
    for (int i = 0; i < A_INS; ++i) {
        const int r = (wid * A_INS + i) * 8 + (lane >> 3);
        ga[i] = A8 + (size_t)min(m0 + r, p.M - 1) * lda8 + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < W_INS; ++i) {
        const int r = (wid * W_INS + i) * 8 + (lane >> 3);
        gw[i] = W8 + (size_t)min(n0 + r, p.N - 1) * ldw8 + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
    }
    auto issue = [&](int buf) __attribute__((always_inline)) {
        unsigned char* base = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < A_INS; ++i) {
            __builtin_amdgcn_global_load_lds((glb_void8_t*)ga[i], (lds_void8_t*)(base + (wid * A_INS + i) * 1024), 16, 0, 0);
            ga[i] += BKB;
        }
#pragma unroll
        for (int i = 0; i < W_INS; ++i) {
            __builtin_amdgcn_global_load_lds((glb_void8_t*)gw[i], (lds_void8_t*)(base + A_BYTES + (wid * W_INS + i) * 1024), 16, 0, 0);
            gw[i] += BKB;
        }
    };
    const int kg = lane >> 4, l15 = lane & 15, fx = l15 >> 1;
    // a lane's 32 operand bytes of row l15: k-bytes 32 kg .. 32 kg + 31 = slots 2 kg and 2 kg + 1 (each XOR-ed with fx)
    const int rdlo = l15 * 128 + (((2 * kg) ^ fx) << 4), rdhi = l15 * 128 + (((2 * kg + 1) ^ fx) << 4);

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BKB;
    issue(0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's pieces of tile kt
        __builtin_amdgcn_s_barrier();                              // everyone's pieces; the other buffer is free again
        asm volatile("" ::: "memory");
        if (kt + 1 < nk) issue((kt + 1) & 1);
        const unsigned char* Ab = smem + (kt & 1) * BUF + (wm * TM) * 128;
        const unsigned char* Wb = smem + (kt & 1) * BUF + A_BYTES + (wn * TN) * 128;
        v8i_t a[MI], b[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const uint4 lo = *reinterpret_cast<const uint4*>(Ab + i * 2048 + rdlo), hi = *reinterpret_cast<const uint4*>(Ab + i * 2048 + rdhi);
            a[i] = v8i_t{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const uint4 lo = *reinterpret_cast<const uint4*>(Wb + j * 2048 + rdlo), hi = *reinterpret_cast<const uint4*>(Wb + j * 2048 + rdhi);
            b[j] = v8i_t{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
        }
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int i = 0; i < MI; ++i)      // operands swapped (W first): a lane owns 4 consecutive n of one row m
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b[j], a[i], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }

    // acc[i][j][e] = sum_k q_a[m][k] q_w[n][k],  m = m0 + wm*TM + i*16 + l15,  n = n0 + wn*TN + j*16 + 4*kg + e  ->  real units
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const float s_m = sa[min(m0 + wm * TM + i * 16 + l15, p.M - 1)];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = min(n0 + wn * TN + j * 16 + 4 * kg, p.N - 4);
            const float4 s_n = *reinterpret_cast<const float4*>(sw + n);
            acc[i][j][0] *= s_m * s_n.x; acc[i][j][1] *= s_m * s_n.y; acc[i][j][2] *= s_m * s_n.z; acc[i][j][3] *= s_m * s_n.w;
        }
    }
    // rank-R LoRA extension in bf16 (real units), fragment-shaped direct loads
    if (p.R > 0) {
        const int nrs = (p.R + 31) >> 5;
        for (int s = 0; s < nrs; ++s) {
            const int kk = s * 32 + kg * 8;
            const int kkc = kk < p.R ? kk : 0;
            const unsigned keep = kk < p.R ? 0xffffffffu : 0u;
            bf16x8 fa[MI], fb[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                uint4 v = *reinterpret_cast<const uint4*>(p.U + (size_t)min(m0 + wm * TM + i * 16 + l15, p.M - 1) * p.ldu + kkc);
                v.x &= keep; v.y &= keep; v.z &= keep; v.w &= keep;
                fa[i] = *reinterpret_cast<bf16x8*>(&v);
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                uint4 v = *reinterpret_cast<const uint4*>(p.Bl + (size_t)min(n0 + wn * TN + j * 16 + l15, p.N - 1) * p.ldbl + kkc);
                v.x &= keep; v.y &= keep; v.z &= keep; v.w &= keep;
                fb[j] = *reinterpret_cast<bf16x8*>(&v);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) Mma<T>::mma(acc[i][j], fb[j], fa[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * TM + i * 16 + l15;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = n0 + wn * TN + j * 16 + 4 * kg;
            if (m < p.M && n < p.N) gemm_epilogue_direct4(p, acc[i][j], m, n);
        }
    }
}

// a: as for cvft_gemm (dtype bf16; A / W / lda / ldw ignored; identity row geometry; no fused side path, no masked
// extension); A8 [M][lda8] / W8 [N][ldw8] e4m3 bytes with per-row scales a_scale [M] / w_scale [N].
extern "C" int cvft_gemm_fp8(const cvft_gemm_args* a, const void* A8, int lda8, const float* a_scale, const void* W8, int ldw8,
                             const float* w_scale, void* stream) {
    CVFT_CHECK_ARG(a && A8 && W8 && a_scale && w_scale && a->C, "cvft_gemm_fp8: null operand");
    CVFT_CHECK_ARG(a->dtype == CVFT_BF16 && a->M > 0 && a->N > 0 && a->K > 0 && a->K % 128 == 0 && a->N % 4 == 0 && lda8 >= a->K &&
                   ldw8 >= a->K && lda8 % 16 == 0 && ldw8 % 16 == 0 && (((uintptr_t)A8 | (uintptr_t)W8) & 15) == 0 &&
                   ((uintptr_t)w_scale & 15) == 0,
                   "cvft_gemm_fp8: bf16 output, K %% 128 == 0, N %% 4 == 0, 16-byte aligned fp8 rows and w_scale");
    CVFT_CHECK_ARG(a->ntaps == 1 && a->tap_off[0] == 0 && a->Tm == a->M && a->Tin == a->M && a->Tout == a->M && a->in_stride == 1 &&
                   a->out_stride == 1 && a->out_off == 0 && !a->in_len && !a->La && a->xdrop_p <= 0.f,
                   "cvft_gemm_fp8: identity row geometry only, no fused side path / masked extension");
    GP<bf16_t> p;
    p.M = a->M; p.N = a->N; p.K = a->K; p.Tm = a->M; p.Tin = a->M; p.Tout = a->M;
    p.in_stride = 1; p.out_stride = 1; p.out_off = 0; p.ntaps = 1;
    for (int i = 0; i < 4; ++i) p.tap_off[i] = 0;
    p.in_len = nullptr; p.out_len = a->out_len;
    p.A = nullptr; p.lda = 0; p.W = nullptr; p.ldw = 0;
    p.U = (const bf16_t*)a->U; p.ldu = a->ldu; p.R = a->U ? a->R : 0; p.Bl = (const bf16_t*)a->Bl; p.ldbl = a->ldbl;
    p.bias = a->bias; p.alpha = a->alpha; p.act = a->act;
    p.preact = (bf16_t*)a->preact; p.ldp = a->ldp; p.dact_src = (const bf16_t*)a->dact_src; p.ldd = a->ldd; p.dact = a->dact;
    p.residual = (const bf16_t*)a->residual; p.ldr = a->ldr; p.C = (bf16_t*)a->C; p.ldc = a->ldc;
    p.La = nullptr; p.ldla = 0; p.lora_scale = 0.f; p.Uout = nullptr; p.fuse = 0; p.direct_epi = 1; p.xcd_nsplit = 1;
    p.bytesA = p.bytesW = p.bytesU = p.bytesB = p.bytesL = 0;
    p.vecA = p.vecW = 1; p.vecU = p.vecB = 1;
    p.xdrop_p = 0.f; p.xdrop_seed = (const long long*)a->xdrop_seed; p.odrop_p = a->odrop_p; p.odrop_site = a->odrop_site; p.row_off = 0;
    CVFT_CHECK_ARG(p.odrop_p <= 0.f || (p.xdrop_seed && cvft_drop_rate_ok(p.odrop_p) && p.ldc == p.N), "cvft_gemm_fp8: output dropout needs a seed, p < 1 and ldc == N");
    for (int i = 0; i < 4; ++i) p.xdrop_sites[i] = 0;
    if (p.R > 0)
        CVFT_CHECK_ARG(p.Bl && p.R % 8 == 0 && p.R <= 64 && p.ldu % 8 == 0 && p.ldbl % 8 == 0 &&
                       (((uintptr_t)p.U | (uintptr_t)p.Bl) & 15) == 0, "cvft_gemm_fp8: LoRA operands: R %% 8 == 0, R <= 64, 16-byte aligned");
    CVFT_CHECK_ARG((p.ldc % 4 == 0) && (((uintptr_t)p.C) & 7) == 0 && (!p.bias || (((uintptr_t)p.bias) & 15) == 0) &&
                   (!p.preact || (p.ldp % 4 == 0 && (((uintptr_t)p.preact) & 7) == 0)) &&
                   (!p.dact_src || (p.ldd % 4 == 0 && (((uintptr_t)p.dact_src) & 7) == 0)) &&
                   (!p.residual || (p.ldr % 4 == 0 && (((uintptr_t)p.residual) & 7) == 0)),
                   "cvft_gemm_fp8: register epilogue needs 8-byte aligned bf16 rows and a 16-byte aligned bias");
    const size_t sm = 2 * (128 + 128) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        attr_set = true;
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fp8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    }
    const long tiles = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    hipLaunchKernelGGL(gemm_fp8_kernel, dim3((unsigned)tiles), dim3(512), sm, (hipStream_t)stream, p, (const unsigned char*)A8, lda8, a_scale,
                       (const unsigned char*)W8, ldw8, w_scale);
    cvft_set_kernel_label("gemm_fp8_kernel<e4m3,128,128,4,2,ns2,regepi>");
    CVFT_LAUNCH_CHECK("cvft_gemm_fp8");
    return 0;
}
