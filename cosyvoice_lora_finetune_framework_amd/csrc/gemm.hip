// gemm.hip -- tap-GEMM with fused rank-r LoRA side path and epilogue; LoRA-gradient
// "TN" accumulation.  MFMA 16x16 tiles (bf16 16x16x32 / exact-fp32 16x16x4), LDS-staged
// k-contiguous operand tiles, register-prefetched next tile (issue-early / write-late).
//
// Replaces (reference): lora.py:64-76, nn.Linear/Conv1d/ConvTranspose1d calls of
// modules.py:60-120 & matcha/models/components/decoder.py:35-158, and their dgrad.
#include <stdlib.h>
#include "gemm_common.h"

// Main loop: the K extent (all taps, then the rank-r LoRA segment) is cut into BK-wide tiles.  D tiles
// are kept in flight in registers (these GEMMs are small and latency-bound: the lever is bytes in
// flight), LDS is double-buffered with ONE barrier per tile.  AL = every operand 16-byte aligned and
// K % VEC == 0: operand chunks are raw buffer loads (hardware range check => no predication branches,
// 32-bit offsets, and a statically known number of outstanding loads so hipcc emits counted vmcnt waits:
// loads past the last tile are issued anyway and return zeros).  !AL: guarded element loads.
// Epilogue: accumulators -> LDS (fp32) -> 16-byte coalesced row segments with the whole
// bias/act/act'/residual/mask chain applied on the vectors.
template <typename T, int BM, int BN, int WM, int WN, int D, bool AL, bool FU>
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
    constexpr bool A_FULL = (BM * CPR) % NT == 0, W_FULL = (BN * CPR) % NT == 0;
    constexpr int CLD = BN + 4;
    typedef Mma<T> MM;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* As = reinterpret_cast<T*>(smem_raw);                 // [2][BM*LD]
    T* Ws = As + 2 * BM * LD;                               // [2][BN*LD]
    T* Ls = Ws + 2 * BN * LD;                               // FU: [2][16*LD] LoRA-A tile
    float* Cs = reinterpret_cast<float*>(smem_raw);         // [BM][CLD] (aliases the operand ring after the loop)

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int tiles_n = (p.N + BN - 1) / BN;
    // XCD-aware remap (blocks are dealt round-robin over the 8 XCDs): give each XCD a contiguous range of
    // tile ids, n fastest, so the n-tiles that share an A row panel -- and the W panel they all read --
    // hit the same 4 MiB L2.  Bijective for any grid size (speed only, never correctness).
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, rm = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + loc;
    }
    const int m0 = (bid / tiles_n) * BM;
    const int n0 = (bid % tiles_n) * BN;

    int a_row[A_IT], a_cc[A_IT], a_b[A_IT], a_t[A_IT], a_len[A_IT];
    bool a_ok[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        int c = tid + i * NT;
        a_row[i] = c / CPR;
        a_cc[i] = c % CPR;
        int m = m0 + a_row[i];
        a_ok[i] = (A_FULL || c < BM * CPR) && (m < p.M);
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
        w_ok[i] = (W_FULL || c < BN * CPR) && (n0 + w_row[i] < p.N);
    }

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk_main = (p.K + BK - 1) / BK;
    const int nk_lora = (p.R > 0 && !FU) ? (p.R + BK - 1) / BK : 0;
    const int n_it = p.ntaps * nk_main + nk_lora;

    uint4 ra[D][A_IT], rw[D][W_IT];
    uint4 rl[D];                          // FU: LoRA-A tile chunk (threads < 16*CPR)
    f32x4 uacc[MI];                       // FU: x_tile . La^T (waves with wn == 0)
#pragma unroll
    for (int i = 0; i < MI; ++i) uacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int l_row = tid / CPR, l_cc = tid % CPR;
    const bool l_ok = FU && tid < 16 * CPR && l_row < p.R;
    __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(FU ? p.La : p.W), 0, FU ? p.bytesL : 0u, 0x00020000);
    const unsigned ol = l_ok ? (unsigned)(((size_t)l_row * p.ldla + l_cc * VEC) * sizeof(T)) : CVFT_OOB;

    // Loader state machine (tiles are requested strictly in order): per-slot offsets are set up once per
    // segment (tap / LoRA), only the k offset advances per tile.
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(p.A), 0, p.bytesA, 0x00020000);
    __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(p.W), 0, p.bytesW, 0x00020000);
    __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(p.U), 0, p.bytesU, 0x00020000);
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(p.Bl), 0, p.bytesB, 0x00020000);
    unsigned oa[A_IT], ow[W_IT];          // AL: byte offsets (CVFT_OOB = invalid row)
    const T* pa[A_IT];                    // !AL: row pointers
    const T* pw[W_IT];
    bool va[A_IT], vw[W_IT];
    int ld_seg = -1, ld_kt = 0, ld_nk = 0, ld_klim = 0;
    auto seg_setup = [&](int seg) __attribute__((always_inline)) {
        if (seg < p.ntaps) {
            const int toff = p.tap_off[seg];
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                int ti = a_t[i] * p.in_stride + toff;
                va[i] = a_ok[i] && ti >= 0 && ti < p.Tin && ti < a_len[i];
                const size_t e = (size_t)(a_b[i] * p.Tin + (va[i] ? ti : 0)) * p.lda + a_cc[i] * VEC;
                oa[i] = va[i] ? (unsigned)(e * sizeof(T)) : CVFT_OOB;
                pa[i] = p.A + e;
            }
#pragma unroll
            for (int i = 0; i < W_IT; ++i) {
                vw[i] = w_ok[i];
                const size_t e = (size_t)(w_ok[i] ? n0 + w_row[i] : 0) * p.ldw + (size_t)seg * p.K + w_cc[i] * VEC;
                ow[i] = vw[i] ? (unsigned)(e * sizeof(T)) : CVFT_OOB;
                pw[i] = p.W + e;
            }
            ld_klim = p.K;
            ld_nk = nk_main;
        } else {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                va[i] = a_ok[i] && p.R > 0 && !FU;
                const size_t e = (size_t)(a_ok[i] ? m0 + a_row[i] : 0) * p.ldu + a_cc[i] * VEC;
                oa[i] = va[i] ? (unsigned)(e * sizeof(T)) : CVFT_OOB;
                pa[i] = p.U + e;
            }
#pragma unroll
            for (int i = 0; i < W_IT; ++i) {
                vw[i] = w_ok[i] && p.R > 0 && !FU;
                const size_t e = (size_t)(w_ok[i] ? n0 + w_row[i] : 0) * p.ldbl + w_cc[i] * VEC;
                ow[i] = vw[i] ? (unsigned)(e * sizeof(T)) : CVFT_OOB;
                pw[i] = p.Bl + e;
            }
            ld_klim = p.R;
            ld_nk = 0x3fffffff;           // terminal segment: later tiles fall beyond ld_klim and load zeros
        }
        ld_seg = seg;
        ld_kt = 0;
    };
    auto gload = [&](uint4 (&ra_)[A_IT], uint4 (&rw_)[W_IT], uint4& rl_) __attribute__((always_inline)) {
        if (ld_kt == ld_nk) seg_setup(ld_seg + 1);
        const int k0 = ld_kt * BK;
        const bool lora = ld_seg >= p.ntaps;
        if (FU && AL) {
            const unsigned off = (!lora && k0 + l_cc * VEC < p.K) ? ol + (unsigned)(k0 * sizeof(T)) : CVFT_OOB;
            u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsL, (int)off, 0, 0);
            rl_ = make_uint4(v[0], v[1], v[2], v[3]);
        }
        if (AL) {
            const __amdgpu_buffer_rsrc_t ra_src = lora ? rsU : rsA;
            const __amdgpu_buffer_rsrc_t rw_src = lora ? rsB : rsW;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const unsigned off = (k0 + a_cc[i] * VEC < ld_klim) ? oa[i] + (unsigned)(k0 * sizeof(T)) : CVFT_OOB;
                u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ra_src, (int)off, 0, 0);
                ra_[i] = make_uint4(v[0], v[1], v[2], v[3]);
            }
#pragma unroll
            for (int i = 0; i < W_IT; ++i) {
                const unsigned off = (k0 + w_cc[i] * VEC < ld_klim) ? ow[i] + (unsigned)(k0 * sizeof(T)) : CVFT_OOB;
                u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw_src, (int)off, 0, 0);
                rw_[i] = make_uint4(v[0], v[1], v[2], v[3]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                T tmp[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const int kk = k0 + a_cc[i] * VEC + e;
                    tmp[e] = (va[i] && kk < ld_klim) ? pa[i][k0 + e] : from_f32<T>(0.f);
                }
                ra_[i] = *reinterpret_cast<uint4*>(tmp);
            }
#pragma unroll
            for (int i = 0; i < W_IT; ++i) {
                T tmp[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const int kk = k0 + w_cc[i] * VEC + e;
                    tmp[e] = (vw[i] && kk < ld_klim) ? pw[i][k0 + e] : from_f32<T>(0.f);
                }
                rw_[i] = *reinterpret_cast<uint4*>(tmp);
            }
        }
        ld_kt += 1;
    };
    auto sstore = [&](int buf, const uint4 (&ra_)[A_IT], const uint4 (&rw_)[W_IT], const uint4& rl_) __attribute__((always_inline)) {
        T* Ab = As + buf * BM * LD;
        T* Wb = Ws + buf * BN * LD;
        if (FU && tid < 16 * CPR) *reinterpret_cast<uint4*>(&Ls[buf * 16 * LD + l_row * LD + MM::chunk_sw(l_row, l_cc) * VEC]) = rl_;
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            if (A_FULL || tid + i * NT < BM * CPR) *reinterpret_cast<uint4*>(&Ab[a_row[i] * LD + MM::chunk_sw(a_row[i], a_cc[i]) * VEC]) = ra_[i];
#pragma unroll
        for (int i = 0; i < W_IT; ++i)
            if (W_FULL || tid + i * NT < BN * CPR) *reinterpret_cast<uint4*>(&Wb[w_row[i] * LD + MM::chunk_sw(w_row[i], w_cc[i]) * VEC]) = rw_[i];
    };
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const T* Ab = As + buf * BM * LD;
        const T* Wb = Ws + buf * BN * LD;
#pragma unroll
        for (int ks = 0; ks < BK; ks += MM::K) {
            typename MM::Frag a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = MM::load_sw(&Ab[(wm * TM + i * 16 + (lane & 15)) * LD + ks], lane);
#pragma unroll
            for (int j = 0; j < NI; ++j) b[j] = MM::load_sw(&Wb[(wn * TN + j * 16 + (lane & 15)) * LD + ks], lane);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) MM::mma(acc[i][j], a[i], b[j]);
            if (FU && wn == 0) {          // wave-uniform: side-path product on the A fragments already in registers
                typename MM::Frag bl = MM::load_sw(&Ls[buf * 16 * LD + (lane & 15) * LD + ks], lane);
#pragma unroll
                for (int i = 0; i < MI; ++i) MM::mma(uacc[i], a[i], bl);
            }
        }
    };

    seg_setup(0);
    // prologue: tiles 0..D-1 in flight, tile 0 staged, its register set refilled with tile D
    static_for<D>([&](auto dc) __attribute__((always_inline)) {
        constexpr int d = decltype(dc)::value;
        gload(ra[d], rw[d], rl[d]);
    });
    sstore(0, ra[0], rw[0], rl[0]);
    gload(ra[0], rw[0], rl[0]);
    __syncthreads();
    // steady state: stage tile it+1 (register set dn) into the other LDS buffer, refill that set with
    // tile it+1+D, compute tile it.  No data-dependent control flow between the loads and their use.
    int it0 = 0;
    for (; it0 + D <= n_it; it0 += D) {
        static_for<D>([&](auto dc) __attribute__((always_inline)) {
            constexpr int d = decltype(dc)::value;
            constexpr int dn = (d + 1) % D;
            const int it = it0 + d;
            sstore((it + 1) & 1, ra[dn], rw[dn], rl[dn]);
            gload(ra[dn], rw[dn], rl[dn]);
            compute(it & 1);
            __syncthreads();
        });
    }
    static_for<D>([&](auto dc) __attribute__((always_inline)) {      // remainder (< D tiles), block-uniform
        constexpr int d = decltype(dc)::value;
        constexpr int dn = (d + 1) % D;
        const int it = it0 + d;
        if (it < n_it) {
            sstore((it + 1) & 1, ra[dn], rw[dn], rl[dn]);
            compute(it & 1);
            __syncthreads();
        }
    });

    if (FU) {
        // rank-R extension with the in-kernel U: stage  s*U (bf16/fp32, zero-padded to BK) as an A tile and the
        // LoRA-B rows of this n-tile as a W tile in LDS buffer 0, run one more k-tile; n-tile-0 blocks publish U.
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            if (A_FULL || tid + i * NT < BM * CPR) *reinterpret_cast<uint4*>(&As[a_row[i] * LD + a_cc[i] * VEC]) = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < W_IT; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            const bool ok = w_ok[i] && w_cc[i] * VEC < p.R;
            if (AL) {
                const unsigned off = ok ? (unsigned)(((size_t)(n0 + w_row[i]) * p.ldbl + w_cc[i] * VEC) * sizeof(T)) : CVFT_OOB;
                u32x4 t4 = __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)off, 0, 0);
                v = make_uint4(t4[0], t4[1], t4[2], t4[3]);
            }
            if (W_FULL || tid + i * NT < BN * CPR) *reinterpret_cast<uint4*>(&Ws[w_row[i] * LD + MM::chunk_sw(w_row[i], w_cc[i]) * VEC]) = v;
        }
        if (tid < 16 * CPR) *reinterpret_cast<uint4*>(&Ls[l_row * LD + l_cc * VEC]) = make_uint4(0, 0, 0, 0);
        __syncthreads();
        if (wn == 0) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wm * TM + i * 16 + (lane >> 4) * 4 + r, col = lane & 15;
                    const T uv = from_f32<T>(uacc[i][r] * p.lora_scale);
                    As[row * LD + MM::chunk_sw(row, col / VEC) * VEC + (col % VEC)] = uv;
                    if (n0 == 0 && p.Uout && m0 + row < p.M && col < p.R) p.Uout[(size_t)(m0 + row) * p.ldu + col] = uv;
                }
        }
        __syncthreads();
        compute(0);
        __syncthreads();
    }

    // ------------------------------------------------------------- epilogue
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Cs[(wm * TM + i * 16 + (lane >> 4) * 4 + r) * CLD + wn * TN + j * 16 + (lane & 15)] = acc[i][j][r];
    __syncthreads();
    gemm_epilogue_store<T, BM, BN, NT>(p, Cs, m0, n0, tid);
}

template <typename T>
static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename T, int BM, int BN, int WM, int WN, int D, bool AL, bool FU = false>
static int gemm_launch_cfg(const GP<T>& p, hipStream_t st) {
    constexpr int VEC = 16 / sizeof(T);
    constexpr int BK = (sizeof(T) == 2) ? 64 : 16;
    constexpr int LD = BK + VEC;
    size_t ring = (size_t)2 * (BM + BN + (FU ? 16 : 0)) * LD * sizeof(T);
    size_t cs = (size_t)BM * (BN + 4) * sizeof(float);
    size_t sm = ring > cs ? ring : cs;
    auto kern = gemm_kernel<T, BM, BN, WM, WN, D, AL, FU>;
    static bool attr_set = false;             // per instantiation; a host call per launch is visible in eager mode
    if (sm > 48 * 1024 && !attr_set) {
        attr_set = true;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) {
            cvft_set_error("cvft_gemm: hipFuncSetAttribute(%zu) failed: %s", sm, hipGetErrorString(e));
            return -2;
        }
    }
    long tiles = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(WM * WN * 64), sm, st, p);
    cvft_set_kernel_label("gemm_kernel<%s,%d,%d,%d,%d>%s", sizeof(T) == 2 ? "bf16" : "f32", BM, BN, WM, WN, FU ? ",fusedU" : "");
    CVFT_LAUNCH_CHECK("cvft_gemm");
    return 0;
}

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
    p.La = (const T*)a->La; p.ldla = a->ldla; p.lora_scale = a->lora_scale; p.Uout = (T*)a->Uout; p.fuse = a->La != nullptr; p.direct_epi = 0; p.xcd_nsplit = 1;
    p.bytesL = 0;
    p.xdrop_p = a->xdrop_p; p.xdrop_seed = (const long long*)a->xdrop_seed;
    for (int i = 0; i < 4; ++i) p.xdrop_sites[i] = a->xdrop_sites[i];
    p.odrop_p = a->odrop_p; p.odrop_site = a->odrop_site; p.row_off = 0;
    if (p.odrop_p > 0.f) {
        if (!p.xdrop_seed || !cvft_drop_rate_ok(p.odrop_p) || p.N % 4 != 0 || p.ldc != p.N || p.Tm != p.M || p.out_stride != 1 || p.out_off != 0 || p.Tout != p.Tm) {
            cvft_set_error("cvft_gemm: output dropout needs a seed, 2^-16 <= p <= 1 - 2^-16, N %% 4 == 0, ldc == N and identity row geometry");
            return -1;
        }
    }
    if (p.fuse) { p.R = a->R; p.U = nullptr; }
    if (p.xdrop_p > 0.f) {
        if (sizeof(T) != 2 || p.fuse || !a->U || !p.xdrop_seed || p.R % 16 != 0 || p.R > 64 || !cvft_drop_rate_ok(p.xdrop_p)) {
            cvft_set_error("cvft_gemm: masked rank extension needs bf16, U / Bl with R %% 16 == 0, R <= 64, a seed and 2^-16 <= p <= 1 - 2^-16");
            return -1;
        }
    }
    p.vecA = (a->K % VEC == 0) && (a->lda % VEC == 0) && al16<T>(a->A);
    p.vecW = (a->K % VEC == 0) && (a->ldw % VEC == 0) && al16<T>(a->W);
    p.vecU = p.R > 0 && (p.R % VEC == 0) && (a->ldu % VEC == 0) && al16<T>(a->U);
    p.vecB = p.R > 0 && (p.R % VEC == 0) && (a->ldbl % VEC == 0) && al16<T>(a->Bl);

    const size_t bA = (size_t)(p.M / p.Tm) * p.Tin * p.lda * sizeof(T), bW = (size_t)p.N * p.ldw * sizeof(T);
    const size_t bU = p.R > 0 ? (size_t)p.M * p.ldu * sizeof(T) : 0, bB = p.R > 0 ? (size_t)p.N * p.ldbl * sizeof(T) : 0;
    const size_t lim = 0x7fff0000u;
    p.bytesA = (unsigned)bA; p.bytesW = (unsigned)bW; p.bytesU = (unsigned)bU; p.bytesB = (unsigned)bB;
    long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    bool al = p.vecA && p.vecW && bA < lim && bW < lim;
    if (p.fuse) {
        const size_t bL = (size_t)p.R * p.ldla * sizeof(T);
        p.bytesL = (unsigned)bL;
        p.bytesU = 0;
        p.vecB = (a->ldbl % VEC == 0) && al16<T>(a->Bl);
        const bool ok = al && p.vecB && (a->ldla % VEC == 0) && al16<T>(a->La) && bL < lim && bB < lim && p.N > 32 &&
                        (!p.Uout || a->ldu >= p.R);
        if (!ok) {
            cvft_set_error("cvft_gemm: fused LoRA side path needs 16-byte aligned operands, K %% %d == 0 and N > 32", VEC);
            return -1;
        }
        long t128f = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
        if constexpr (sizeof(T) == 2) {
            static const int gcfg = getenv("CVFT_GLDS_CFG") ? atoi(getenv("CVFT_GLDS_CFG")) : 0;
            if (gcfg >= 0) {
                int rc = gemm_glds_launch(p, st, gcfg);
                if (rc != 1) return rc;
            }
        }
        if (p.R > 16) {
            cvft_set_error("cvft_gemm: fused LoRA side path with 16 < R <= 48 needs the bf16 LDS-DMA kernel (K %% 64 == 0, identity rows)");
            return -1;
        }
        if constexpr (sizeof(T) == 2) {
            if (p.K >= 2048 && t128f >= 256) return gemm_launch_cfg<T, 256, 128, 4, 2, 2, true, true>(p, st);
        }
        return gemm_launch_cfg<T, 64, 64, 2, 2, 4, true, true>(p, st);
    }
    al = al && (p.R == 0 || (p.vecU && p.vecB)) && bU < lim && bB < lim;
    if (al) {
        if constexpr (sizeof(T) == 2) {
            // LDS-DMA kernels take every eligible shape (CVFT_GLDS_CFG: -1 = off; experiments)
            static const int gcfg = getenv("CVFT_GLDS_CFG") ? atoi(getenv("CVFT_GLDS_CFG")) : 0;
            if (gcfg >= 0 && p.N <= 64 && p.xdrop_p <= 0.f) {
                int rc = skinny_launch(p, st);
                if (rc != 1) return rc;
            }
            if (gcfg >= 0) {
                int rc = gemm_glds_launch(p, st, gcfg);
                if (rc != 1) return rc;
            }
        }
        if (p.xdrop_p > 0.f) {
            cvft_set_error("cvft_gemm: masked rank extension: launch not eligible for the LDS-DMA register-epilogue kernels");
            return -1;
        }
        if (p.N <= 32) return gemm_launch_cfg<T, 32, 32, 2, 1, 4, true>(p, st);
        // measured on MI355X (tools/bench_kernels.py, CVFT_GEMM_CFG sweep): 64x64 tiles with 4 k-tiles in flight win
        // on every step shape up to K = 1024; long-K GEMMs (w_2 dgrad / forward, K = 4096) prefer 256x128 x 8 waves.
        if constexpr (sizeof(T) == 2) {
            if (p.ntaps * p.K >= 2048 && t128 >= 256) return gemm_launch_cfg<T, 256, 128, 4, 2, 2, true>(p, st);
        }
        return gemm_launch_cfg<T, 64, 64, 2, 2, 4, true>(p, st);
    }
    if (p.xdrop_p > 0.f) {
        cvft_set_error("cvft_gemm: masked rank extension needs 16-byte aligned operands");
        return -1;
    }
    return gemm_launch_cfg<T, 64, 64, 2, 2, 2, false>(p, st);     // unaligned / odd-K operands: generic element loads
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
    if (a->La) {
        CVFT_CHECK_ARG(a->Bl && a->R > 0 && a->R <= 48 && a->ldla >= a->K && a->ldbl >= a->R, "cvft_gemm: fused LoRA needs 0 < R <= 48");
        CVFT_CHECK_ARG(a->ntaps == 1 && a->Tm == a->Tin && a->in_stride == 1 && a->tap_off[0] == 0 && a->out_stride == 1 &&
                       a->out_off == 0 && a->Tout == a->Tin, "cvft_gemm: fused LoRA side path needs identity row geometry");
    } else if (a->U) {
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

// v3 (aligned operands, r == 16 * k): one block = 64 columns x (4 waves x RW rows).  Each wave issues ALL
// its 16-byte row-segment loads up front (8 lanes cover a 64-column segment of one row, 8 rows per
// wave-instruction), accumulates every rank for its rows in registers, the 4 waves are combined through
// LDS and each output element gets ONE fp32 atomic per block, issued as 256-byte-contiguous wave-instructions.
template <typename T, int RW>
__global__ void __launch_bounds__(256) lora_rank_accum_vec_kernel(int M, int Cn, int r, const T* __restrict__ Wd, int ldw,
                                                                   const T* __restrict__ Rk, int ldr,
                                                                   float* __restrict__ out, int ldo, int transpose_out,
                                                                   size_t part_stride, int iters) {
    // part_stride != 0: deterministic two-stage mode -- block row y writes its slab to out + y*part_stride with
    // plain stores (no atomics; cvft_lora_grad_reduce sums the slabs).  part_stride == 0: fp32 atomics into out.
    constexpr int VEC = 16 / sizeof(T);
    constexpr int CG = 64 / VEC;        // lanes per 64-column row segment
    constexpr int RG = 64 / CG;         // rows per wave-instruction
    constexpr int NL = RW / RG;         // load instructions per wave
    extern __shared__ float red[];      // [4 waves][RG row groups][16][64]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g = lane / CG;
    const int c = blockIdx.x * 64 + (lane % CG) * VEC;
    const bool cok = c < Cn;
    for (int j0 = 0; j0 < r; j0 += 16) {
        float acc[16][VEC];
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[j][e] = 0.f;
        // the block walks `iters` slabs of 4*RW rows; per slab every lane has NL 16-byte loads in flight
        for (int itr = 0; itr < iters; ++itr) {
            const int row0 = (blockIdx.y * iters + itr) * (4 * RW) + w * RW + g;
            uint4 wv[NL];
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                const int row = row0 + u * RG;
                wv[u] = (cok && row < M) ? *reinterpret_cast<const uint4*>(Wd + (size_t)row * ldw + c) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                const int row = row0 + u * RG;
                T rk[16];
                if (row < M) {
#pragma unroll
                    for (int q = 0; q < 16 / VEC; ++q)
                        *reinterpret_cast<uint4*>(&rk[q * VEC]) = *reinterpret_cast<const uint4*>(Rk + (size_t)row * ldr + j0 + q * VEC);
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j) rk[j] = from_f32<T>(0.f);
                }
                const T* we = reinterpret_cast<const T*>(&wv[u]);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float rv = to_f32(rk[j]);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[j][e] += rv * to_f32(we[e]);
                }
            }
        }
        // combine the RG row groups x 4 waves through LDS (wide ds_write, conflict-free column reads); a register
        // xor-shuffle tree here cost ~18 us of exposed ds_bpermute latency per launch
        __syncthreads();
        float* mine = red + (size_t)((w * RG + g) * 16) * 64 + (lane % CG) * VEC;
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int q = 0; q < VEC / 4; ++q)
                *reinterpret_cast<float4*>(mine + j * 64 + q * 4) = make_float4(acc[j][q * 4], acc[j][q * 4 + 1], acc[j][q * 4 + 2], acc[j][q * 4 + 3]);
        __syncthreads();
        for (int idx = threadIdx.x; idx < 16 * 64; idx += 256) {
            const int j = idx >> 6, col = idx & 63;
            const int cl = blockIdx.x * 64 + col;
            if (cl >= Cn) continue;
            float v = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < 4 * RG; ++s2) v += red[(size_t)(s2 * 16 + j) * 64 + col];
            float* base = out + (size_t)blockIdx.y * part_stride;
            float* dst = transpose_out ? &base[(size_t)cl * ldo + j0 + j] : &base[(size_t)(j0 + j) * ldo + cl];
            if (part_stride) *dst = v; else atomicAdd(dst, v);
        }
    }
}

template <typename T>
static int rank_accum_launch(int M, int Cn, int r, const void* Wd, int ldw, const void* Rk, int ldr, float* out, int ldo,
                             int transpose_out, hipStream_t st, int part_rows = 0) {
    // part_rows: rows per slab block (multiple of 64).  64 / 128 -> one pass of RW = 16 / 32 rows per wave;
    // multiples of 256 -> RW = 64 with part_rows / 256 passes.
    constexpr int VEC = 16 / sizeof(T);
    const bool vec = (Cn % VEC == 0) && (ldw % VEC == 0) && ((reinterpret_cast<uintptr_t>(Wd) & 15) == 0) &&
                     (r % 16 == 0) && (ldr % VEC == 0) && ((reinterpret_cast<uintptr_t>(Rk) & 15) == 0);
    int colblocks = (Cn + 63) / 64;
    if (part_rows && !vec) return 2;
    if (vec) {
        // rows per block 64 / 128 / 256: the largest that still gives >= 512 blocks (or the caller's choice)
        int RW = 64;
        while (RW > 16 && (long)colblocks * ((M + 4 * RW - 1) / (4 * RW)) < 512) RW >>= 1;
        int iters = 1;
        if (part_rows) {
            if (part_rows >= 256) { RW = 64; iters = part_rows / 256; } else RW = part_rows / 4;
        }
        const size_t part_stride = part_rows ? (size_t)r * Cn : 0;
        dim3 grid(colblocks, (M + 4 * RW * iters - 1) / (4 * RW * iters));
        const size_t smr = (size_t)4 * (64 / (64 / VEC)) * 16 * 64 * sizeof(float);       // 4 waves x RG groups x 16 x 64
#define RV_LAUNCH(RWv)                                                                                                    \
    do {                                                                                                                  \
        auto kern = lora_rank_accum_vec_kernel<T, RWv>;                                                                   \
        if (smr > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smr); \
        hipLaunchKernelGGL(kern, grid, dim3(256), smr, st, M, Cn, r, (const T*)Wd, ldw, (const T*)Rk, ldr, out, ldo,       \
                           transpose_out, part_stride, iters);                                                           \
    } while (0)
        if (RW == 64) RV_LAUNCH(64); else if (RW == 32) RV_LAUNCH(32); else RV_LAUNCH(16);
#undef RV_LAUNCH
        return 0;
    }
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

int lora_rank_mfma_launch(int M, int Cn, int r, const void* Wd, int ldw, const void* Rk, int ldr, float* part,
                          int transpose_out, int rows_per_block, hipStream_t st);

// Two-stage (deterministic, atomic-free) form: slabs, then ONE reduce launch for every adapter of the step.
extern "C" int cvft_lora_rank_partial(int dtype, int M, int Cn, int r, const void* Wd, int ldw, const void* Rk, int ldr,
                                      float* part, int transpose_out, int rows_per_block, void* stream) {
    CVFT_CHECK_ARG(dtype == CVFT_F32 || dtype == CVFT_BF16, "cvft_lora_rank_partial: bad dtype");
    CVFT_CHECK_ARG(M > 0 && Cn > 0 && r > 0 && r % 16 == 0 && ldw >= Cn && ldr >= r && Wd && Rk && part, "cvft_lora_rank_partial: bad args");
    CVFT_CHECK_ARG(rows_per_block > 0 && rows_per_block % 32 == 0, "cvft_lora_rank_partial: rows_per_block must be a multiple of 32");
    hipStream_t st = (hipStream_t)stream;
    int ldo = transpose_out ? r : Cn;
    if (dtype == CVFT_BF16) {          // matrix-core kernel (lora_grad.hip) for every eligible bf16 launch
        static const int off = getenv("CVFT_RANK_VALU") ? atoi(getenv("CVFT_RANK_VALU")) : 0;
        if (!off && lora_rank_mfma_launch(M, Cn, r, Wd, ldw, Rk, ldr, part, transpose_out, rows_per_block, st) == 0) {
            CVFT_LAUNCH_CHECK("cvft_lora_rank_partial");
            return 0;
        }
    }
    CVFT_CHECK_ARG(rows_per_block == 64 || rows_per_block == 128 || (rows_per_block >= 256 && rows_per_block % 256 == 0),
                   "cvft_lora_rank_partial: the VALU path needs rows_per_block 64, 128 or a multiple of 256");
    int rc = dtype == CVFT_F32 ? rank_accum_launch<float>(M, Cn, r, Wd, ldw, Rk, ldr, part, ldo, transpose_out, st, rows_per_block)
                               : rank_accum_launch<bf16_t>(M, Cn, r, Wd, ldw, Rk, ldr, part, ldo, transpose_out, st, rows_per_block);
    CVFT_CHECK_ARG(rc == 0, "cvft_lora_rank_partial: operands must be 16-byte aligned with C %% VEC == 0");
    CVFT_LAUNCH_CHECK("cvft_lora_rank_partial");
    return 0;
}

// tasks[t] = {slab base, grad base, rows, cols, slab_pitch, slab_stride, nsplit, 0} (int64 x 8):
//   grad[i*cols + j] += sum_s slab[s*slab_stride + i*slab_pitch + j], fixed order.
__global__ void __launch_bounds__(256) lora_grad_reduce_kernel(const long long* __restrict__ tasks) {
    const long long* t = tasks + (size_t)blockIdx.y * 8;
    const float* slab = reinterpret_cast<const float*>(t[0]);
    float* grad = reinterpret_cast<float*>(t[1]);
    const long long rows = t[2], cols = t[3], pitch = t[4], stride = t[5];
    const int nsplit = (int)t[6];
    const long long numel = rows * cols;
    // 16-byte form: four consecutive elements of one row per thread, four slabs in flight (fixed summation order)
    const bool vec = (cols % 4 == 0) && (pitch % 4 == 0) && (stride % 4 == 0) &&
                     (((reinterpret_cast<uintptr_t>(slab) | reinterpret_cast<uintptr_t>(grad)) & 15) == 0);
    if (vec) {
        for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < numel; i += (long long)gridDim.x * 1024) {
            const long long src = (pitch == cols) ? i : (i / cols) * pitch + (i % cols);
            const float* sp = slab + src;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            int k = 0;
            for (; k + 4 <= nsplit; k += 4) {
                const float4 a = *reinterpret_cast<const float4*>(sp + (size_t)k * stride);
                const float4 b = *reinterpret_cast<const float4*>(sp + (size_t)(k + 1) * stride);
                const float4 c = *reinterpret_cast<const float4*>(sp + (size_t)(k + 2) * stride);
                const float4 d = *reinterpret_cast<const float4*>(sp + (size_t)(k + 3) * stride);
                s.x += (a.x + b.x) + (c.x + d.x); s.y += (a.y + b.y) + (c.y + d.y);
                s.z += (a.z + b.z) + (c.z + d.z); s.w += (a.w + b.w) + (c.w + d.w);
            }
            for (; k < nsplit; ++k) {
                const float4 a = *reinterpret_cast<const float4*>(sp + (size_t)k * stride);
                s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            }
            float4 g = *reinterpret_cast<float4*>(grad + i);
            g.x += s.x; g.y += s.y; g.z += s.z; g.w += s.w;
            *reinterpret_cast<float4*>(grad + i) = g;
        }
        return;
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < numel; i += (long long)gridDim.x * 256) {
        const long long src = (pitch == cols) ? i : (i / cols) * pitch + (i % cols);
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += slab[(size_t)k * stride + src];
        grad[i] += s;
    }
}
extern "C" int cvft_lora_grad_reduce(int ntasks, const void* tasks, int max_blocks_x, void* stream) {
    CVFT_CHECK_ARG(ntasks >= 0 && tasks && max_blocks_x > 0, "cvft_lora_grad_reduce: bad args");
    if (ntasks == 0) return 0;
    hipLaunchKernelGGL(lora_grad_reduce_kernel, dim3(max_blocks_x, ntasks), dim3(256), 0, (hipStream_t)stream, (const long long*)tasks);
    CVFT_LAUNCH_CHECK("cvft_lora_grad_reduce");
    return 0;
}

// ------------------------------------------------------------------------------
// LoRA shadows: for every [rows x cols] fp32 master in the flat parameter buffer write a compute-dtype
// (bf16) copy at the same offset of `flat_c` and a TRANSPOSED copy ([cols x rows]) at the same offset
// of `flat_t` -- one launch per optimiser step for all adapters (32x32 LDS tiles).
//   tiles[t] = {param offset, rows, cols, tile_row * 65536 + tile_col}
// ------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) lora_shadow_kernel(const long long* __restrict__ tiles, const float* __restrict__ flat_p) {
    __shared__ float tl[32][33];
    const long long* d = tiles + (size_t)blockIdx.x * 6;
    const float* src = flat_p + d[0];
    bf16_t* dc = reinterpret_cast<bf16_t*>(d[1]);
    bf16_t* dtp = reinterpret_cast<bf16_t*>(d[2]);
    const int rows = (int)(d[3] & 0xffffffffLL), cols = (int)(d[3] >> 32);
    const int tr = (int)(d[4] & 0xffffffffLL) * 32, tc = (int)(d[4] >> 32) * 32;
    const int pc = (int)(d[5] & 0xffffffffLL), pt = (int)(d[5] >> 32);
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;     // 32 x 8
    for (int i = ly; i < 32; i += 8) {
        int rr = tr + i, cc = tc + lx;
        float v = 0.f;
        if (rr < rows && cc < cols) {
            v = src[(size_t)rr * cols + cc];
            dc[(size_t)rr * pc + cc] = (bf16_t)v;
        }
        tl[i][lx] = v;
    }
    __syncthreads();
    for (int i = ly; i < 32; i += 8) {
        int cc = tc + i, rr = tr + lx;
        if (rr < rows && cc < cols) dtp[(size_t)cc * pt + rr] = (bf16_t)tl[lx][i];
    }
}
extern "C" int cvft_lora_shadow(int ntiles, const void* tiles, const float* flat_p, void* stream) {
    CVFT_CHECK_ARG(ntiles >= 0 && tiles && flat_p, "cvft_lora_shadow: bad args");
    if (ntiles == 0) return 0;
    hipLaunchKernelGGL(lora_shadow_kernel, dim3(ntiles), dim3(256), 0, (hipStream_t)stream, (const long long*)tiles, flat_p);
    CVFT_LAUNCH_CHECK("cvft_lora_shadow");
    return 0;
}
