// gemm_glds.hip -- bf16 tap-GEMM for the common identity-geometry case (Linear / LoRALinear forward and dgrad),
// operand tiles moved HBM/L2 -> LDS by the LDS-DMA path (global_load_lds_dwordx4, cdna_hip_programming.md
// section 5): no staging registers, no ds_write pass, the next k-tile lands while the current one feeds the MFMAs.
//
//   C[M,N] = epilogue( alpha * ( A[M,K] . W[N,K]^T  +  U[M,R] . Bl[N,R]^T ) )          K % 64 == 0, R % 8 == 0
//
// LDS image of one k-tile: unpadded 128-byte rows (64 bf16), rows in tile order, so one wave-instruction
// (64 lanes x 16 B) fills 8 consecutive rows -- the DMA's "wave-uniform base + lane * 16" rule.  Bank conflicts of
// the ds_read_b128 fragment reads are removed by an XOR on the SOURCE side: slot s of row r holds global chunk
// s ^ ((r >> 1) & 7) in the A image, s ^ glds_wswz(r) in the W image (whose fragments read permuted rows, see the
// column map below); readers apply the same XOR.  NS LDS buffers, one barrier per k-tile.
// The rank-R LoRA extension runs on fragment-shaped direct loads (16 rows x 16 B per lane group) issued before the
// main loop.  Everything else (row geometry, taps, masks, fp32, odd shapes) stays on gemm.hip's register-staged kernel.
//
// Replaces (reference): lora.py:64-76 and the nn.Linear calls of the estimator / encoders, and their dgrad.
#include <stdlib.h>
#include "gemm_common.h"


// RT > 0 (fused side path): U = lora_scale * A . La^T  (La [R <= 16*RT][K]) is produced inside the launch: the La k-tile
// rides along as 16*RT more DMA rows, the wn == 0 waves run RT extra MFMAs per A fragment, the bf16 result goes through
// a small LDS panel into every wave's extension fragments, and the n-tile-0 blocks publish it to Uout for backward.
// NS = LDS stages: tiles kt+1 .. kt+NS-1 are in flight while tile kt feeds the MFMAs (counted s_waitcnt vmcnt, raw
// s_barrier -- __syncthreads() would drain the DMA queue).  These GEMMs are small (1-4 blocks per CU, 4-16 k-tiles):
// with one tile of lookahead every k-tile cost a full memory round trip (~0.9 us measured in the training step).
__device__ unsigned long long cvft_glds_stamps[2 * 16 * 8];      // diagnostics (CVFT_GLDS_STAMP=1, tools/glds_stamps.py)

template <int BM, int BN, int WM, int WN, int RT, int NS, bool DE, bool STAMP = false, bool DX = false>
__global__ void __launch_bounds__(WM * WN * 64) gemm_glds_kernel(GP<bf16_t> p) {
    typedef bf16_t T;
    constexpr int NW = WM * WN, NT = NW * 64, BK = 64;
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 16, NI = TN / 16;
    constexpr bool FU = RT > 0;
    constexpr int RP = 16 * RT;                                   // padded rank of the fused side path
    constexpr int L_PIECES = 2 * RT;                              // 8-row DMA pieces of the La tile, dealt round-robin to the waves
    constexpr int LP_MAX = (L_PIECES + NW - 1) / NW;
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, L_BYTES = RP * 128, BUF = A_BYTES + W_BYTES + L_BYTES;
    constexpr int UP_OFF = NS * BUF;                              // FU: [BM][RP] bf16 panel behind the ring
    // 8-row DMA pieces of the A / W tile, dealt round-robin to the waves (piece = wid + i * NW).  RAGGED: the piece count
    // is not a multiple of the wave count (96-row tiles on 12 waves) -- the last round is guarded (wave-uniform) and
    // the vmcnt immediates below are picked per wave
    constexpr int A_PIECES = BM / 8, W_PIECES = BN / 8;
    constexpr int A_INS = (A_PIECES + NW - 1) / NW, W_INS = (W_PIECES + NW - 1) / NW;      // DMA wave-instructions per wave per k-tile (max)
    constexpr bool RAGGED = (A_PIECES % NW != 0) || (W_PIECES % NW != 0);
    static_assert(!(RAGGED && FU), "ragged piece deal: plain kernels only");
    constexpr int INS = A_INS + W_INS;                            // DMA instructions per wave per tile (+ its La pieces)
    static_assert(NS >= 2 && (NS - 2) * (INS + LP_MAX) < 64, "vmcnt is a 6-bit counter");
    static_assert(LP_MAX <= 2, "wait dispatch below handles 0, 1 or 2 La pieces per wave");
    constexpr int CLD = BN + 4;
    static_assert(BM % 8 == 0 && BN % 8 == 0, "tile rows must split into 8-row DMA pieces");
    static_assert(TM % 16 == 0 && TN % 32 == 0, "wave tile: 16-row x 32-column units (column map pairs the n tiles)");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* Cs = reinterpret_cast<float*>(smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int tiles_n = (p.N + BN - 1) / BN;
    int bid = blockIdx.x;
    int m0, n0;
    if (p.xcd_nsplit > 1) {
        // Wide-N launches (W larger than an XCD's 4 MB L2): with linear tile ranges every XCD streams ALL of W once per
        // row panel (PMC: 325 MB of L2 misses for 19 MB of operands at 5328x4096x1024).  Give each XCD a rectangle --
        // xcd_nsplit column groups x 8/xcd_nsplit row groups -- so its W slice stays L2-resident while it walks its
        // row panels (m slow, n fast).  Workgroups are dealt round-robin to the XCDs (id & 7); the grid is padded to
        // 8 x the largest rectangle and surplus blocks leave before any barrier.
        const int tiles_m = (p.M + BM - 1) / BM;
        const int cx = p.xcd_nsplit, cy = 8 / cx;
        const int xcd = bid & 7, loc = bid >> 3;
        const int tn_per = (tiles_n + cx - 1) / cx, tm_per = (tiles_m + cy - 1) / cy;
        const int tn0 = (xcd % cx) * tn_per, tm0 = (xcd / cx) * tm_per;
        const int rn = min(tn_per, tiles_n - tn0), rmm = min(tm_per, tiles_m - tm0);
        if (rn <= 0 || rmm <= 0 || loc >= rn * rmm) return;
        m0 = (tm0 + loc / rn) * BM;
        n0 = (tn0 + loc % rn) * BN;
    } else {
        // XCD-aware bijective remap (see gemm.hip): contiguous tile ranges per XCD, n fastest
        const int nwg = gridDim.x, q = nwg >> 3, rm = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + loc;
        m0 = (bid / tiles_n) * BM;
        n0 = (bid % tiles_n) * BN;
    }

    // DMA sources: lane -> (row piece*8 + lane/8, slot lane%8); rows past the edge are clamped (their outputs are
    // never stored), the chunk is XOR-ed so the LDS image comes out swizzled
    const char* ga[A_INS];
    const char* gw[W_INS];
#pragma unroll
    for (int i = 0; i < A_INS; ++i) {
        const int r = (wid + i * NW) * 8 + (lane >> 3);
        const int gc = (lane & 7) ^ ((r >> 1) & 7);
        const int m = min(m0 + r, p.M - 1);
        ga[i] = reinterpret_cast<const char*>(p.A + (size_t)m * p.lda + gc * 8);
    }
#pragma unroll
    for (int i = 0; i < W_INS; ++i) {
        const int r = (wid + i * NW) * 8 + (lane >> 3);
        const int gc = (lane & 7) ^ glds_wswz(r);
        const int n = min(n0 + r, p.N - 1);
        gw[i] = reinterpret_cast<const char*>(p.W + (size_t)n * p.ldw + gc * 8);
    }
    const char* gl[LP_MAX > 0 ? LP_MAX : 1];                      // FU: wave w moves La pieces w, w + NW, ...
    const int my_lp = FU ? (wid < L_PIECES ? (L_PIECES - wid + NW - 1) / NW : 0) : 0;     // wave-uniform
    if (FU) {
#pragma unroll
        for (int i = 0; i < LP_MAX; ++i) {
            const int r = (wid + i * NW) * 8 + (lane >> 3);
            const int gc = (lane & 7) ^ ((r >> 1) & 7);
            gl[i] = reinterpret_cast<const char*>(p.La + (size_t)min(r, p.R - 1) * p.ldla + gc * 8);
        }
    }
    auto issue = [&](int buf) __attribute__((always_inline)) {
        unsigned char* base = smem + buf * BUF;
        if (FU) {
#pragma unroll
            for (int i = 0; i < LP_MAX; ++i)
                if (i < my_lp) {
                    __builtin_amdgcn_global_load_lds((glb_void_t*)gl[i], (lds_void_t*)(base + A_BYTES + W_BYTES + (wid + i * NW) * 1024), 16, 0, 0);
                    gl[i] += BK * 2;
                }
        }
#pragma unroll
        for (int i = 0; i < A_INS; ++i) {
            if (A_PIECES % NW == 0 || wid + i * NW < A_PIECES) {       // wave-uniform
                __builtin_amdgcn_global_load_lds((glb_void_t*)ga[i], (lds_void_t*)(base + (wid + i * NW) * 1024), 16, 0, 0);
                ga[i] += BK * 2;
            }
        }
#pragma unroll
        for (int i = 0; i < W_INS; ++i) {
            if (W_PIECES % NW == 0 || wid + i * NW < W_PIECES) {
                __builtin_amdgcn_global_load_lds((glb_void_t*)gw[i], (lds_void_t*)(base + A_BYTES + (wid + i * NW) * 1024), 16, 0, 0);
                gw[i] += BK * 2;
            }
        }
    };

    // rank-R extension operands, fragment-shaped, straight to registers (in flight during the whole main loop)
    constexpr int RS = 2;                         // up to R = 64
    uint4 ua[RS][MI], ub[RS][NI];                 // raw 16-byte chunks; masked and reinterpreted only after the main loop
    const int kg = lane >> 4, l15 = lane & 15;
    const int nrs = (p.R + 31) >> 5;
#pragma unroll
    for (int s = 0; s < RS; ++s) {
#pragma unroll
        for (int i = 0; i < MI; ++i) ua[s][i] = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NI; ++j) ub[s][j] = make_uint4(0, 0, 0, 0);
    }
    if (p.R > 0) {
        // unconditional loads from clamped addresses; the lane mask (k >= R -> 0) is applied where the fragments are
        // consumed, after the main loop: a guarded load makes hipcc branch around it and wait vmcnt(0) at the join, a
        // mask applied here makes it wait right here -- either way one exposed memory round trip before the first DMA
#pragma unroll
        for (int s = 0; s < RS; ++s) {
            const int kk = s * 32 + kg * 8;
            const int kkc = kk < p.R ? kk : 0;
            if (s * 32 < p.R) {                                   // wave-uniform
                if (!FU) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        const int m = min(m0 + wm * TM + i * 16 + l15, p.M - 1);
                        ua[s][i] = *reinterpret_cast<const uint4*>(p.U + (size_t)m * p.ldu + kkc);
                    }
                }
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int n = min(n0 + wn * TN + glds_col(j, l15), p.N - 1);
                    ub[s][j] = *reinterpret_cast<const uint4*>(p.Bl + (size_t)n * p.ldbl + kkc);
                }
            }
        }
    }

    f32x4 acc[MI][NI];
    f32x4 uacc[MI][RT > 0 ? RT : 1];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int t = 0; t < (RT > 0 ? RT : 1); ++t) uacc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // fragment read addresses: row l15 of a 16-row group, chunk (ks*4 + kg) ^ (l15 >> 1)
    const int fx = l15 >> 1;
    const int rd0 = l15 * 128 + ((kg ^ fx) << 4);
    const int rd1 = l15 * 128 + (((4 + kg) ^ fx) << 4);
    // W side (column map): slot l15 of tile j is image row glds_col(j, l15); its XOR is lane-constant
    const int wrow = 8 * (l15 >> 2) + (l15 & 3), fw = ((l15 >> 1) & 1) | ((l15 >> 2) << 1);
    const int rw0 = wrow * 128 + ((kg ^ fw) << 4);
    const int rw1 = wrow * 128 + (((4 + kg) ^ fw) << 4);
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const unsigned char* Ab = smem + buf * BUF + (wm * TM) * 128;
        const unsigned char* Wb = smem + buf * BUF + A_BYTES + (wn * TN) * 128;
        // all fragment reads of the k-tile are issued before its first MFMA: the second k-step's reads complete under
        // the first k-step's MFMAs (hipcc places the partial lgkmcnt waits).
        // Measured round 2 (tools/bench_cfg.py, 5328 x {4096x1024, 3072x1024, 1024x4096, 1024x1024}): s_setprio(1) around
        // this MFMA block 603 -> 392 TF/s (the MFMA waves then starve their partners' DMA issue and LDS reads: the
        // kernel is bound by memory-op issue and LDS bandwidth -- 12 KB of fragment reads per wave and k-tile, 192 KB per
        // CU and round against 1 024 MFMA cycles -- not by the matrix core); fragment reads hoisted above the DMA issue
        // 603 -> 385; s_setprio(2) on the DMA-issue segment instead: no change.
        bf16x8 a[2][MI], b[2][NI];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int rd = ks ? rd1 : rd0, rw = ks ? rw1 : rw0;
#pragma unroll
            for (int i = 0; i < MI; ++i) a[ks][i] = *reinterpret_cast<const bf16x8*>(Ab + i * 2048 + rd);
#pragma unroll
            for (int j = 0; j < NI; ++j) b[ks][j] = *reinterpret_cast<const bf16x8*>(Wb + glds_col(j, 0) * 128 + rw);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int rd = ks ? rd1 : rd0;
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i) Mma<T>::mma(acc[i][j], b[ks][j], a[ks][i]);     // swapped: lane owns 4 consecutive n
            if (FU && wn == 0) {                                  // wave-uniform
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    const bf16x8 la = *reinterpret_cast<const bf16x8*>(smem + buf * BUF + A_BYTES + W_BYTES + t * 2048 + rd);
#pragma unroll
                    for (int i = 0; i < MI; ++i) Mma<T>::mma(uacc[i][t], a[ks][i], la);
                }
            }
        }
    };

    const int nk = p.K / BK;
    // RAGGED: pieces this wave does NOT move in its last deal round (0, 1 or 2; wave-uniform)
    const int short_by = (A_PIECES % NW != 0 && wid + (A_INS - 1) * NW >= A_PIECES ? 1 : 0) +
                         (W_PIECES % NW != 0 && wid + (W_INS - 1) * NW >= W_PIECES ? 1 : 0);
#pragma unroll
    for (int t = 0; t < NS - 1; ++t)
        if (t < nk) issue(t);
    int cb = 0, ib = NS - 1;                                    // buffer of tile kt / of tile kt + NS - 1
    for (int kt = 0; kt < nk; ++kt) {
        const bool rec = STAMP && blockIdx.x == 100 && (wid == 0 || wid == NW - 1) && lane == 0 && kt >= 4 && kt < 12;
        unsigned long long* sb = cvft_glds_stamps + (wid == 0 ? 0 : 128) + (kt - 4) * 8;
        if (rec) sb[0] = __builtin_readcyclecounter();
        // this wave's pieces of tile kt have landed once at most the NS-2 younger tiles are still outstanding
        // (loads retire in order; the LoRA fragment loads are older than every tile)
        if (RAGGED && kt + NS - 2 < nk) {
            if (short_by == 0) wait_vmcnt<(NS - 2) * INS>();
            else if (short_by == 1) wait_vmcnt<(NS - 2) * (INS > 1 ? INS - 1 : 0)>();
            else wait_vmcnt<(NS - 2) * (INS > 2 ? INS - 2 : 0)>();
        } else if (kt + NS - 2 < nk) {
            if (my_lp == 2) wait_vmcnt<(NS - 2) * (INS + 2)>();
            else if (my_lp == 1) wait_vmcnt<(NS - 2) * (INS + 1)>();
            else wait_vmcnt<(NS - 2) * INS>();
        } else {
            wait_vmcnt<0>();                                    // pipeline tail
        }
        if (rec) sb[1] = __builtin_readcyclecounter();
        __builtin_amdgcn_s_barrier();                           // everyone's pieces; and buffer ib (tile kt-1) is free
        asm volatile("" ::: "memory");
        if (rec) sb[2] = __builtin_readcyclecounter();
        if (kt + NS - 1 < nk) issue(ib);
        if (rec) sb[3] = __builtin_readcyclecounter();
        compute(cb);
        if (rec) { asm volatile("s_nop 0" ::: "memory"); sb[4] = __builtin_readcyclecounter(); }
        cb = (cb + 1 == NS) ? 0 : cb + 1;
        ib = (ib + 1 == NS) ? 0 : ib + 1;
    }
    if (FU) {
        bf16_t* Up = reinterpret_cast<bf16_t*>(smem + UP_OFF);
        if (wn == 0) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = wm * TM + i * 16 + kg * 4 + r, col = t * 16 + l15;
                        const bf16_t uv = from_f32<T>(col < p.R ? uacc[i][t][r] * p.lora_scale : 0.f);
                        Up[row * RP + col] = uv;
                        if (n0 == 0 && p.Uout && m0 + row < p.M && col < p.R) p.Uout[(size_t)(m0 + row) * p.ldu + col] = uv;
                    }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < RS; ++s) {
            const int kk = s * 32 + kg * 8;
            if (kk < RP) {
#pragma unroll
                for (int i = 0; i < MI; ++i) ua[s][i] = *reinterpret_cast<const uint4*>(Up + (wm * TM + i * 16 + l15) * RP + kk);
            }
        }
    }
    if constexpr (DX) {
        // masked rank extension (lora_dropout dgrad): per 16-wide rank tile t, acc += mask_t / (1 - p) * (U_t . Bl_t^T), the
        // mask taken over the OUTPUT elements -- a lane owns 4 consecutive n of one row, exactly one keep4 group
        const unsigned thr = cvft_drop_thr(p.xdrop_p);
        const float inv = 1.f / (1.f - p.xdrop_p);
        const int ntile = p.R >> 4;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (t < ntile) {                                      // wave-uniform
                constexpr int dummy = 0;
                (void)dummy;
                const int s = t >> 1;
                const unsigned keep = ((kg >> 1) == (t & 1)) ? 0xffffffffu : 0u;     // this tile's 16 k of the 32-wide step
                bf16x8 fa[MI], fb[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    uint4 v = ua[s][i];
                    v.x &= keep; v.y &= keep; v.z &= keep; v.w &= keep;
                    fa[i] = *reinterpret_cast<bf16x8*>(&v);
                }
#pragma unroll
                for (int j = 0; j < NI; ++j) fb[j] = *reinterpret_cast<bf16x8*>(&ub[s][j]);
                const unsigned long long key = cvft_drop_key(p.xdrop_seed, p.xdrop_sites[t]);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const unsigned long long m = (unsigned long long)(min(m0 + wm * TM + i * 16 + l15, p.M - 1) + p.row_off);
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        f32x4 a2 = f32x4{0.f, 0.f, 0.f, 0.f};
                        Mma<T>::mma(a2, fb[j], fa[i]);
                        const int n = min(n0 + wn * TN + glds_col(j, 4 * kg), p.N - 4);
                        bool k4[4];
                        cvft_keep4(key, (m * (unsigned long long)p.N + n) >> 2, thr, k4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j][e] += k4[e] ? a2[e] * inv : 0.f;
                    }
                }
            }
    } else {
#pragma unroll
    for (int s = 0; s < RS; ++s)
        if (s < nrs) {
            const unsigned keep = (s * 32 + kg * 8) < p.R ? 0xffffffffu : 0u;
            bf16x8 fa[MI], fb[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                uint4 v = ua[s][i];
                v.x &= keep; v.y &= keep; v.z &= keep; v.w &= keep;
                fa[i] = *reinterpret_cast<bf16x8*>(&v);
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                uint4 v = ub[s][j];
                v.x &= keep; v.y &= keep; v.z &= keep; v.w &= keep;
                fb[j] = *reinterpret_cast<bf16x8*>(&v);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) Mma<T>::mma(acc[i][j], fb[j], fa[i]);
        }
    }

    // acc[i][j][r] = C[m0 + wm*TM + i*16 + l15][n0 + wn*TN + glds_col(j, 4*kg) + r]
    // DE (register epilogue) is a separate instantiation: with both epilogues in one kernel the LDS path lost ~6 %
    // (accumulators left the AGPRs, twice the code)
    if constexpr (DE) {
        const bool wide = glds_wide_epilogue(p);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = m0 + wm * TM + i * 16 + l15;
#pragma unroll
            for (int jp = 0; jp < NI / 2; ++jp) {
                const int n = n0 + wn * TN + 32 * jp + 8 * kg;           // = glds_col(2 * jp, 4 * kg); tile 2*jp+1 holds n+4 .. n+7
                if (m >= p.M) continue;
                if (wide) {                                               // block-uniform
                    if (n < p.N) gemm_epilogue_direct8(p, acc[i][2 * jp], acc[i][2 * jp + 1], m, n);
                } else {
                    if (n < p.N) gemm_epilogue_direct4(p, acc[i][2 * jp], m, n);
                    if (n + 4 < p.N) gemm_epilogue_direct4(p, acc[i][2 * jp + 1], m, n + 4);
                }
            }
        }
    } else {
        __syncthreads();                                        // operand ring is dead: reuse it for the fp32 tile
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)    // one ds_write_b128 per tile: rows 272 B apart -> 16 lanes hit 16 distinct 16-byte slots
                *reinterpret_cast<float4*>(&Cs[(wm * TM + i * 16 + l15) * CLD + wn * TN + glds_col(j, 4 * kg)]) =
                    make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        __syncthreads();
        gemm_epilogue_store<T, BM, BN, NT>(p, Cs, m0, n0, tid);
    }
}

template <int BM, int BN, int WM, int WN, int RT, int NS, bool DE, bool DX = false>
static int glds_launch_de(const GP<bf16_t>& p, hipStream_t st) {
    constexpr bool FU = RT > 0;
    size_t ring = (size_t)NS * (BM + BN + 16 * RT) * 128 + (size_t)BM * 32 * RT;
    size_t cs = DE ? 0 : (size_t)BM * (BN + 4) * sizeof(float);        // fp32 staging tile only for the LDS epilogue
    size_t sm = ring > cs ? ring : cs;
    auto kern = gemm_glds_kernel<BM, BN, WM, WN, RT, NS, DE, false, DX>;
    static bool attr_set = false;             // per instantiation; a host call per launch is visible in eager mode
    if (sm > 48 * 1024 && !attr_set) {
        attr_set = true;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) {
            cvft_set_error("cvft_gemm: hipFuncSetAttribute(%zu) failed: %s", sm, hipGetErrorString(e));
            return -2;
        }
    }
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    long tiles = (long)tiles_m * tiles_n;
    GP<bf16_t> q = p;
    q.xcd_nsplit = 1;
    {   // rectangle mapping when W does not fit an XCD's L2 and the split balances (see kernel)
        static const int rect_on = getenv("CVFT_GLDS_RECT") ? atoi(getenv("CVFT_GLDS_RECT")) : 1;
        const size_t wbytes = (size_t)p.N * p.K * 2;
        if (rect_on && wbytes > (size_t)3 << 20) {
            int cx = 2;
            while (cx < 8 && wbytes / cx > ((size_t)2 << 20) + ((size_t)1 << 18)) cx *= 2;
            const int cy = 8 / cx;
            const long tn_per = (tiles_n + cx - 1) / cx, tm_per = (tiles_m + cy - 1) / cy;
            const long padded = 8 * tn_per * tm_per;
            if (tiles_m >= cy && padded * 100 <= tiles * 108) {          // <= 8 % idle slots
                q.xcd_nsplit = cx;
                tiles = padded;
            }
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(WM * WN * 64), sm, st, q);
    if (FU) cvft_set_kernel_label("gemm_glds_kernel<bf16,%d,%d,%d,%d,ns%d,%s>,fusedU%d", BM, BN, WM, WN, NS, DE ? "regepi" : "ldsepi", 16 * RT);
    else cvft_set_kernel_label("gemm_glds_kernel<bf16,%d,%d,%d,%d,ns%d,%s>%s", BM, BN, WM, WN, NS, DE ? "regepi" : "ldsepi", DX ? ",xdrop" : "");
    CVFT_LAUNCH_CHECK("cvft_gemm");
    return 0;
}

template <int BM, int BN, int WM, int WN, int RT = 0, int NS = 2>
static int glds_launch_cfg(const GP<bf16_t>& p, hipStream_t st) {
    if (p.xdrop_p > 0.f) {          // masked rank extension: register epilogue, non-fused instantiations only
        if constexpr (RT == 0) {
            if (glds_direct_epilogue(p)) return glds_launch_de<BM, BN, WM, WN, 0, NS, true, true>(p, st);
        }
        return 1;
    }
    if (glds_direct_epilogue(p)) return glds_launch_de<BM, BN, WM, WN, RT, NS, true>(p, st);
    return glds_launch_de<BM, BN, WM, WN, RT, NS, false>(p, st);
}

// Returns 1 when the launch is not eligible (the caller falls back to gemm.hip's register-staged kernel).
int gemm_glds_launch(const GP<bf16_t>& p_in, hipStream_t st, int cfg) {
    static const int direct_mode = getenv("CVFT_GLDS_DIRECT") ? atoi(getenv("CVFT_GLDS_DIRECT")) : 2;   // 0 off, 1 only plain epilogues, 2 whenever legal
    if (cfg != -256) {   // LLM-sized launches: 256x256 tile on the 8-phase pipeline (gemm_p256.hip; -256 = its own remainder launch)
        const int rc = gemm_p256_launch(p_in, st);
        if (rc != 1) return rc;
    }
    GP<bf16_t> p = p_in;
    const bool simple = !p.preact && !p.dact_src && !p.residual;
    p.direct_epi = direct_mode == 2 || (direct_mode == 1 && simple);
    const bool ident = p.ntaps == 1 && p.tap_off[0] == 0 && p.in_stride == 1 && p.Tin == p.Tm && !p.in_len;
    if (!ident || p.K % 64 != 0 || !p.vecA || !p.vecW || p.N <= 32) return 1;
    const long t64 = (long)((p.M + 63) / 64) * ((p.N + 63) / 64);
    // Stage count: a deep DMA pipeline only pays while the CU still holds every block that wants to run there
    // (16 / 24 KB per stage), measured on cold operands (tools/bench_cold.py): <= 2 blocks per CU and >= 4 k-tiles ->
    // 4 stages (64x64) / 3 (128x64); a third block per CU -> 3; beyond that 2.  CVFT_GLDS_NS overrides (experiments).
    static const int ns_env = getenv("CVFT_GLDS_NS") ? atoi(getenv("CVFT_GLDS_NS")) : 0;
    static const long big_thr = getenv("CVFT_GLDS_BIG_T64") ? atol(getenv("CVFT_GLDS_BIG_T64")) : 1000;
    const bool big = t64 >= big_thr;
    const long blocks = big ? (long)((p.M + 127) / 128) * ((p.N + 63) / 64) : t64;
    const int nk = p.K / 64;
    int ns = 2;
    if (nk >= 4) {
        if (big) ns = blocks <= 512 ? 3 : 2;
        else ns = blocks <= 512 ? 4 : (blocks <= 768 ? 3 : 2);
    }
    if (ns_env) ns = ns_env;
    if (p.fuse) {     // gemm_launch has already checked the La / Bl alignment
        if (p.R < 1 || p.R > 48 || p.Tm != p.M || p.out_stride != 1 || p.out_off != 0) return 1;
        if (p.R > 16) {                                          // stacked q|k|v adapters: three rank tiles
            if (p.R % 8 != 0) return 1;
            if (big) {
                if (ns >= 3) return glds_launch_cfg<128, 64, 4, 2, 3, 3>(p, st);
                return glds_launch_cfg<128, 64, 4, 2, 3, 2>(p, st);
            }
            if (ns >= 4) return glds_launch_cfg<64, 64, 2, 2, 3, 4>(p, st);
            if (ns == 3) return glds_launch_cfg<64, 64, 2, 2, 3, 3>(p, st);
            return glds_launch_cfg<64, 64, 2, 2, 3, 2>(p, st);
        }
        if (big) {
            if (ns >= 3) return glds_launch_cfg<128, 64, 4, 2, 1, 3>(p, st);
            return glds_launch_cfg<128, 64, 4, 2, 1, 2>(p, st);
        }
        if (ns >= 4) return glds_launch_cfg<64, 64, 2, 2, 1, 4>(p, st);
        if (ns == 3) return glds_launch_cfg<64, 64, 2, 2, 1, 3>(p, st);
        return glds_launch_cfg<64, 64, 2, 2, 1, 2>(p, st);
    }
    if (p.R > 0 && (p.R % 8 != 0 || p.R > 64 || !p.vecU || !p.vecB)) return 1;
    // measured on MI355X (tools/sweep_gemm.py): 128x64 x 8 waves once there are >= 4 64x64 tiles per CU, else 64x64
    static const int big_env = getenv("CVFT_GLDS_BIG") ? atoi(getenv("CVFT_GLDS_BIG")) : 0;     // experiment hook
    if (big && big_env && nk >= 8 && t64 >= 1024) {
        switch (big_env) {
            case 1: return glds_launch_cfg<128, 128, 2, 2, 0, 3>(p, st);
            case 3: return glds_launch_cfg<128, 128, 2, 4, 0, 3>(p, st);
            case 4: return glds_launch_cfg<256, 128, 4, 2, 0, 2>(p, st);
            case 6: return glds_launch_cfg<128, 128, 2, 2, 0, 2>(p, st);
            case 8: return glds_launch_cfg<128, 128, 2, 4, 0, 2>(p, st);
            case 17: return glds_launch_cfg<128, 256, 4, 4, 0, 3>(p, st);
            case 19: return glds_launch_cfg<96, 256, 3, 4, 0, 3>(p, st);
            case 13: return glds_launch_cfg<128, 128, 4, 2, 0, 3>(p, st);
            case 14: return glds_launch_cfg<128, 128, 4, 2, 0, 4>(p, st);
            case 15: {          // stamped build of the default 128x128 kernel (register epilogue)
                auto kern = gemm_glds_kernel<128, 128, 4, 2, 0, 2, true, true>;
                hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
                GP<bf16_t> q = p;
                q.xcd_nsplit = 1;
                hipLaunchKernelGGL(kern, dim3((unsigned)(((p.M + 127) / 128) * ((p.N + 127) / 128))), dim3(512), 65536, st, q);
                cvft_set_kernel_label("gemm_glds_kernel<bf16,128,128,4,2,ns2,regepi,stamped>");
                return 0;
            }
            case 11: if (ns >= 3) return glds_launch_cfg<128, 64, 4, 2, 0, 3>(p, st);
                     return glds_launch_cfg<128, 64, 4, 2, 0, 2>(p, st);
            default: break;
        }
    }
    // LLM-sized launches with a deep k-loop: 128x128 tile, 8 waves as 4x2 (32x64 wave tiles), 2 stages = 64 KB, two
    // blocks per CU.  Per k-tile a wave issues 12 fragment reads + 4 DMA pieces for 16 MFMAs instead of 8 + 3 for 8
    // (in-kernel stamps of the retired 256x256 prototype: ~23 cycles per ds_read_b128 and ~60 per DMA piece against 16 per MFMA -- the
    // 32x32 wave tile is issue-bound), and a byte from L2 feeds 64 FLOP instead of 43.  Measured cold (tools/bench_cfg.py):
    // 5328x4096x1024 541 -> 616 TFLOP/s, 5328x3072x1024 514 -> 613, 5328x1024x4096 581 -> 604, 5328x1024x1024 equal;
    // 4 waves (64x64 wave tiles) or 192-row / 192-column tiles at one block per CU are slower.
    // In-kernel stamps of this configuration (CVFT_GLDS_BIG=15, tools/glds_stamps.py; 5328x4096x1024, L2-warm): a k-tile takes
    // ~1900 cycles per wave -- 400-700 waiting for its DMA pieces (issued one iteration earlier: ~1500 cycles of latency
    // at ~9.5 TB/s of aggregate L2 -> LDS traffic), ~100 in the barrier, 360-530 issuing 4 pieces, 680 for 12 fragment
    // reads + 16 MFMAs (256 of them matrix-core time).  A third / fourth stage (one block per CU) is slower (468 vs 602
    // TFLOP/s): what is missing is FLOP per L2 byte, i.e. a 256-wide tile (retired prototype: DESIGN.md section 9), not pipeline depth.  (Issuing the
    // fragment reads before the DMA pieces, so that the issue time covers the LDS latency, regressed to 365 TFLOP/s.)
    // Less than one round of 128x128 blocks (N = 1024 at M = 5328: 336 blocks on 512 slots, 80 CUs hold two): a 256-wide
    // tile with 3 stages at ONE block per CU -- 96x256 on 12 waves (224 blocks) or 128x256 on 16 -- same 32x64 wave tiles,
    // 25 % less L2 -> LDS traffic per FLOP, two k-tiles in flight.  Measured cold (tools/bench_cfg.py), 128x128 -> 128x256 ->
    // 96x256: 5328x1024x4096 631 -> 696 -> 709 TFLOP/s, x1024x3072 584 -> 638 -> 652, x1024x1024 485 -> 507 -> 501; on the
    // multi-round shapes (N = 3072 / 4096) 128x256 is equal and 96x256 10-15 % slower, so they keep two 128x128 blocks per CU.
    // ... when this chain has the chip to itself: the 96x256 tile owns its CU (12 waves, 135 KB of LDS), so with other chains in
    // flight (joint mode: LLM + 2 x Flow) their workgroups wait for whole CUs to drain; two co-residing 128x128 blocks per CU are
    // slower alone and faster in the step (same-box A/B, joint B = 16: 23.62 -> 23.16 ms, 22.70 on a second box).  CVFT_GLDS_WIDE=0/1 forces.
    static const int wide_env = getenv("CVFT_GLDS_WIDE") ? atoi(getenv("CVFT_GLDS_WIDE")) : -1;
    const int wide_on = wide_env >= 0 ? wide_env : (cvft_concurrent_chains() >= 3 ? 0 : 1);
    if (big && wide_on && nk >= 8 && t64 >= 1000 && p.N >= 256 && (long)((p.M + 95) / 96) * ((p.N + 255) / 256) <= 256)
        return glds_launch_cfg<96, 256, 3, 4, 0, 3>(p, st);
    if (big && (nk >= 8 || big_env == 12) && t64 >= 1000) return glds_launch_cfg<128, 128, 4, 2, 0, 2>(p, st);
    if (big) {
        if (ns >= 3) return glds_launch_cfg<128, 64, 4, 2, 0, 3>(p, st);
        return glds_launch_cfg<128, 64, 4, 2, 0, 2>(p, st);
    }
    // 64x64 x 4 waves stays the best small tile: 128x64 / 64x128 with 4 waves (64x32 wave tiles) measured 10-40 % slower on
    // the estimator / encoder shapes (M = 4000..4640, N = 256..1536), where block count matters more than issue efficiency
    if (ns >= 4) return glds_launch_cfg<64, 64, 2, 2, 0, 4>(p, st);
    if (ns == 3) return glds_launch_cfg<64, 64, 2, 2, 0, 3>(p, st);
    return glds_launch_cfg<64, 64, 2, 2, 0, 2>(p, st);
}

extern "C" int cvft_debug_glds_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(cvft_glds_stamps), sizeof(unsigned long long) * 2 * 16 * 8);
}
