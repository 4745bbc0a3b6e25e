// gemm_big.hip -- bf16 GEMM for the LLM-sized linears (M ~ 5000, N >= 2048): 256 x 256 output tile, 8 wavefronts as
// 2 (M) x 4 (N), each owning a 128 x 64 sub-tile (32 accumulator tiles), operands HBM/L2 -> LDS by LDS-DMA.
//
//   C[M,N] = epilogue( alpha * ( A[M,K] . W[N,K]^T  +  U[M,R] . Bl[N,R]^T ) )          K % 64 == 0, R % 8 == 0, R <= 64
//
// Why a second kernel beside gemm_glds.hip: its 128 x 64 tile moves 1 byte L2 -> LDS per 43 FLOP and keeps at most
// one k-tile per block in flight; at ~1 us of L2 latency that caps it near 650 TFLOP/s on L2-resident operands (540 in
// the training step).  A 256 x 256 tile needs 1 byte per 128 FLOP, so the 64 KB a CU can hold in flight carry 3x the
// math.  At one block per CU nothing hides a wave's own load phase, so the two wave rows run STAGGERED by one barrier:
//
//        slot      4kt         4kt+1       4kt+2       4kt+3       4kt+4
//        row 0     L(kt,0)     M(kt,0)     L(kt,1)     M(kt,1)     L(kt+1,0)
//        row 1     M(kt-1,1)   L(kt,0)     M(kt,0)     L(kt,1)     M(kt,1)
//
//   L(kt,ks): 12 ds_read_b128 (the wave's 8 A and 4 W fragments of k-step ks of k-tile kt) + its share of the DMA
//   issues, then lgkmcnt(0);  M(kt,ks): 32 MFMAs (and vmcnt(0) after M(kt,1)).  A barrier closes every slot.  Each
//   SIMD holds one wave of each row, so its matrix core always has a wave in M while the other one loads.  (Whole
//   k-tiles per slot would need 96 fragment registers beside the 128 accumulators: spills in the k-loop.)
//
// LDS: two stages of [A rows 0..127 | A rows 128..255 | W rows 0..255] x 128-byte rows (64 KB each), 16-byte slots
// XOR-swizzled on the source side as in gemm_glds.hip.  Buffer life cycle (stage = kt & 1):
//   A-half 0 of tile kt is last read by row 0 in slot 4kt+2          -> refilled (tile kt+2) by row 1 in slot 4kt+3
//   A-half 1 and W of tile kt are last read by row 1 in slot 4kt+3   -> refilled (tile kt+2) by row 0 in slots 4kt+4 / +6
//   row 0 waits for its refills (vmcnt(0)) at the end of slot 4kt+7, row 1 at the end of slot 4kt+4: always a barrier
//   between the wait and the first reader (W of tile kt+2: row 0 in slot 4kt+8).
//
// STATUS (round 1): correct (same results as gemm_glds.hip on every shape tried), NOT yet faster, therefore opt-in
// (CVFT_GEMM_BIG=1 / 2).  Measured on MI355X at 5328 x 3072 x 1024, hot operands: 86 us (gemm_glds: 59 us, hipBLASLt
// 33 us).  By ablation: launch + prologue 13.5 us; k-loop 44 us; register epilogue 27 us (32 MB of 8-byte stores with
// nothing left to overlap them).  In-kernel cycle stamps (CVFT_BIG_STAMP=1, tools/big_stamps.py; one wave per row,
// k-tiles 8..11) show where the k-loop's ~3800 cycles per k-tile go -- ideal is 2048 (2 waves x 64 MFMAs x 16):
//   * an M phase (32 MFMAs) issues in ~600 cycles beside the partner's L phase;
//   * an L phase is 12 ds_read_b128 = ~280 cycles PLUS ~60 cycles per LDS-DMA issue: 530-620 with 4-6 pieces, so the
//     phases are balanced only while a wave issues <= 4 pieces per L phase (row 0 issues 6 + 6, row 1 0 + 4);
//   * vmcnt(0) at the end of M(kt,1) waits 340-480 cycles: pieces issued one slot earlier need ~1100 cycles to land --
//     the refills should target tile kt+2 in the stage just consumed (W quarter wc by wave (1,wc) right after its own
//     last read, A halves by row 0), 4 pieces per L phase and counted vmcnt(4), which gives every piece >= 2 slots;
//   * each s_barrier costs ~80 cycles (4 per k-tile).
// With those fixes the loop should reach ~2700 cycles per k-tile (31 us); the fixed 40 us (prologue + epilogue at one
// block per CU, nothing to overlap them with) then dominate at K = 1024: the epilogue needs 16-byte stores through the
// (by then idle) LDS, and the tile loop needs to be persistent so that one tile's epilogue overlaps the next tile's
// prologue.  With K = 4096 the marginal k-tile already costs 1.0 us (2 PFLOP/s).
//
// The rank-R LoRA extension is applied after the main loop from fragment-shaped direct loads; the epilogue is the
// register epilogue of gemm_common.h (swapped MFMA operands: a lane owns 4 consecutive output columns).
//
// Replaces (reference): lora.py:64-76 / nn.Linear of the LLM's attention and feed-forward projections, and their dgrad.
#include <stdlib.h>
#include "gemm_common.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

#define BIG_STAGE 65536
#define BIG_W_OFF 32768

__device__ unsigned long long cvft_big_stamps[2 * 16 * 8];

template <bool STAMP>
__global__ void __launch_bounds__(512) gemm_big_kernel(GP<bf16_t> p) {
    typedef bf16_t T;
    constexpr int BM = 256, BN = 256, BK = 64, MI = 8, NI = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;      // waves w and w+4 share a SIMD (measured: the other pairings are 25 % slower)
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    int m0, n0;
    {
        int bid = blockIdx.x;
        if (p.xcd_nsplit > 1) {      // XCD rectangles (see gemm_glds.hip): each XCD keeps its W slice in its L2
            const int cx = p.xcd_nsplit, cy = 8 / cx;
            const int xcd = bid & 7, loc = bid >> 3;
            const int tn_per = (tiles_n + cx - 1) / cx, tm_per = (tiles_m + cy - 1) / cy;
            const int tn0 = (xcd % cx) * tn_per, tm0 = (xcd / cx) * tm_per;
            const int rn = min(tn_per, tiles_n - tn0), rmm = min(tm_per, tiles_m - tm0);
            if (rn <= 0 || rmm <= 0 || loc >= rn * rmm) return;      // whole block, before any barrier
            m0 = (tm0 + loc / rn) * BM;
            n0 = (tn0 + loc % rn) * BN;
        } else {
            const int nwg = gridDim.x, q = nwg >> 3, rm = nwg & 7, xcd = bid & 7, loc = bid >> 3;
            bid = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + loc;
            m0 = (bid / tiles_n) * BM;
            n0 = (bid % tiles_n) * BN;
        }
    }

    // DMA piece = 8 rows x 128 B (one wave-instruction): lane -> row lane/8, slot lane%8 holding global chunk slot ^ f(row),
    // f(row) = (row >> 1) & 7 = (prow >> 1) | (piece & 1) << 2.  Addresses are kept as ONE uniform 64-bit base per piece
    // (SALU) plus one of two 32-bit lane offsets per operand (even / odd piece) -- per-piece 64-bit lane pointers would
    // cost 2 VGPRs x 28 pieces and spill into the k-loop.
    const int prow = lane >> 3;
    const char* Ab = reinterpret_cast<const char*>(p.A);
    const char* Wb = reinterpret_cast<const char*>(p.W);
    const unsigned a_pitch = (unsigned)p.lda * 2u, w_pitch = (unsigned)p.ldw * 2u;
    const unsigned sl_e = (unsigned)(((lane & 7) ^ (prow >> 1)) << 4), sl_o = (unsigned)(((lane & 7) ^ ((prow >> 1) | 4)) << 4);
    const unsigned la_e = prow * a_pitch + sl_e, la_o = prow * a_pitch + sl_o;
    const unsigned lw_e = prow * w_pitch + sl_e, lw_o = prow * w_pitch + sl_o;
    // M % 8 == 0 and N % 8 == 0 (launcher): a piece is either wholly inside or wholly past the edge (then skipped: rows
    // past M / N are never stored and may hold anything)
    auto dma_a = [&](int stage, int half, int piece, int kt) __attribute__((always_inline)) {      // piece 0..15 of an A half (uniform)
        const int r0 = m0 + half * 128 + piece * 8;
        if (r0 < p.M) {
            const char* g = Ab + ((size_t)r0 * a_pitch + (size_t)kt * (BK * 2));
            __builtin_amdgcn_global_load_lds((glb_void_t*)(g + ((piece & 1) ? la_o : la_e)),
                                             (lds_void_t*)(smem + stage * BIG_STAGE + (half * 128 + piece * 8) * 128), 16, 0, 0);
        }
    };
    auto dma_w = [&](int stage, int piece, int kt) __attribute__((always_inline)) {                // piece 0..31 (uniform)
        const int r0 = n0 + piece * 8;
        if (r0 < p.N) {
            const char* g = Wb + ((size_t)r0 * w_pitch + (size_t)kt * (BK * 2));
            __builtin_amdgcn_global_load_lds((glb_void_t*)(g + ((piece & 1) ? lw_o : lw_e)),
                                             (lds_void_t*)(smem + stage * BIG_STAGE + BIG_W_OFF + piece * 1024), 16, 0, 0);
        }
    };

    const int nk = p.K / BK;
    const int kg = lane >> 4, l15 = lane & 15;

    // prologue: tile 0 (all of it) and A-half 0 of tile 1, by everyone
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        dma_a(0, 0, wid * 2 + i, 0);
        dma_a(0, 1, wid * 2 + i, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_w(0, wid * 4 + i, 0);
    if (nk > 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i) dma_a(1, 0, wid * 2 + i, 1);
    }

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets inside a 16-row group: row l15, slot (ks*4 + kg) ^ (l15 >> 1)
    const int fx = l15 >> 1;
    const int rd0 = l15 * 128 + ((kg ^ fx) << 4);
    const int rd1 = l15 * 128 + (((4 + kg) ^ fx) << 4);
    const int a_off = wr * 16384, w_off = BIG_W_OFF + wc * 8192;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();                    // the stagger: row 1 runs one slot behind row 0
    __builtin_amdgcn_sched_barrier(0);

    for (int kt = 0; kt < nk; ++kt) {
        const int st = kt & 1;
        const unsigned char* As = smem + st * BIG_STAGE + a_off;
        const unsigned char* Ws = smem + st * BIG_STAGE + w_off;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bool rec = STAMP && blockIdx.x == 8 && wc == 0 && lane == 0 && kt >= 8 && kt < 12;
            unsigned long long* sb = cvft_big_stamps + wr * 128 + ((kt - 8) * 2 + ks) * 8;
            if (rec) sb[0] = __builtin_readcyclecounter();
            // ---------------- L(kt, ks): this wave's 8 A and 4 W fragments of k-step ks (+ its DMA issues)
            const int rd = ks ? rd1 : rd0;
            bf16x8 a[MI], b[NI];
#pragma unroll
            for (int j = 0; j < NI; ++j) b[j] = *reinterpret_cast<const bf16x8*>(Ws + j * 2048 + rd);
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const bf16x8*>(As + i * 2048 + rd);
            {
                if (wr == 0) {                                    // wave-uniform
                    if (kt + 1 < nk) {                            // A-half 1 and W of tile kt+1 -> the other stage (free since row 1's L(kt-1, 1))
#pragma unroll
                        for (int i = 0; i < 2; ++i) dma_a(st ^ 1, 1, wc * 4 + ks * 2 + i, kt + 1);
#pragma unroll
                        for (int i = 0; i < 4; ++i) dma_w(st ^ 1, wc * 8 + ks * 4 + i, kt + 1);
                    }
                } else if (ks == 1) {
                    if (kt + 2 < nk) {                            // A-half 0 of tile kt+2 -> this stage (row 0 finished it in L(kt, 1), a slot ago)
#pragma unroll
                        for (int i = 0; i < 4; ++i) dma_a(st, 0, wc * 4 + i, kt + 2);
                    }
                }
            }
            if (rec) sb[1] = __builtin_readcyclecounter();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // fragments are in registers
            __builtin_amdgcn_sched_barrier(0);
            if (rec) sb[2] = __builtin_readcyclecounter();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (rec) sb[3] = __builtin_readcyclecounter();
            // ---------------- M(kt, ks)
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    Mma<T>::mma(acc[i][j], b[j], a[i]);           // swapped: lane owns 4 consecutive n
                }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (rec) sb[4] = __builtin_readcyclecounter();
            if (ks == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's refills have landed ...
            __builtin_amdgcn_sched_barrier(0);
            if (rec) sb[5] = __builtin_readcyclecounter();
            __builtin_amdgcn_s_barrier();                         // ... and are ordered for every reader from the next slot on
            __builtin_amdgcn_sched_barrier(0);
            if (rec) sb[6] = __builtin_readcyclecounter();
        }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();                    // row 0 makes up for the stagger barrier

    // rank-R extension (LoRA side path as extra k-steps), fragment-shaped direct loads
    if (p.R > 0) {
        const int nrs = (p.R + 31) >> 5;
        for (int s = 0; s < nrs; ++s) {
            const int kk = s * 32 + kg * 8;
            const int kkc = kk < p.R ? kk : 0;
            const unsigned keep = kk < p.R ? 0xffffffffu : 0u;
            bf16x8 fa[MI], fb[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = min(m0 + wr * 128 + i * 16 + l15, p.M - 1);
                uint4 v = *reinterpret_cast<const uint4*>(p.U + (size_t)m * p.ldu + kkc);
                v.x &= keep; v.y &= keep; v.z &= keep; v.w &= keep;
                fa[i] = *reinterpret_cast<bf16x8*>(&v);
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int n = min(n0 + wc * 64 + j * 16 + l15, p.N - 1);
                uint4 v = *reinterpret_cast<const uint4*>(p.Bl + (size_t)n * p.ldbl + kkc);
                v.x &= keep; v.y &= keep; v.z &= keep; v.w &= keep;
                fb[j] = *reinterpret_cast<bf16x8*>(&v);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) Mma<T>::mma(acc[i][j], fb[j], fa[i]);
        }
    }

    // acc[i][j][r] = C[m0 + wr*128 + i*16 + l15][n0 + wc*64 + j*16 + 4*kg + r].  The register epilogue is a large piece
    // of code (bias / activation / activation' / residual / mask variants): inlined for all 32 tiles it was ~200 KB of
    // instructions streamed once per wave through the instruction cache -- 56 us of a 110 us launch.  So: a real loop
    // over the 8 row groups that always stores tile row 0 and then rotates the accumulator rows down by one (28
    // register-to-register moves per step; runtime-indexed accumulators would go to scratch).
#pragma unroll 1
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wr * 128 + i * 16 + l15;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = n0 + wc * 64 + j * 16 + 4 * kg;
            if (m < p.M && n < p.N) gemm_epilogue_direct4(p, acc[0][j], m, n);
        }
#pragma unroll
        for (int k = 0; k + 1 < MI; ++k)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[k][j] = acc[k + 1][j];
    }
}

// Returns 1 when the launch is not eligible (the caller continues with gemm_glds.hip's kernels).
int gemm_big_launch(const GP<bf16_t>& p, hipStream_t st) {
    constexpr int BM = 256, BN = 256;
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    long tiles = (long)tiles_m * tiles_n;
    static const int mode = getenv("CVFT_GEMM_BIG") ? atoi(getenv("CVFT_GEMM_BIG")) : 0;      // 0 off (default: see header), 1 auto, 2 whenever legal
    if (!mode || p.fuse || p.M % 8 != 0 || p.N % 8 != 0 || p.K % 64 != 0 || p.K < 256 || p.R > 64 || p.R % 8 != 0 || (p.R > 0 && (!p.vecU || !p.vecB))) return 1;
    if (mode == 1 && (tiles < 200 || p.N < 2048 || p.M < 2048)) return 1;      // needs ~one block per CU to pay
    GP<bf16_t> q = p;
    q.xcd_nsplit = 1;
    {
        const size_t wbytes = (size_t)p.N * p.K * 2;
        if (wbytes > (size_t)3 << 20) {
            int cx = 2;
            while (cx < 8 && wbytes / cx > ((size_t)2 << 20) + ((size_t)1 << 18)) cx *= 2;
            const int cy = 8 / cx;
            const long tn_per = (tiles_n + cx - 1) / cx, tm_per = (tiles_m + cy - 1) / cy;
            const long padded = 8 * tn_per * tm_per;
            if (tiles_m >= cy && tiles_n >= cx && padded * 100 <= tiles * 108) {
                q.xcd_nsplit = cx;
                tiles = padded;
            }
        }
    }
    const size_t sm = 2 * BIG_STAGE;
    static bool attr_set = false;
    if (!attr_set) {
        attr_set = true;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_big_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_big_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) {
            cvft_set_error("cvft_gemm: hipFuncSetAttribute(%zu) failed: %s", sm, hipGetErrorString(e));
            return -2;
        }
    }
    static const int stamp = getenv("CVFT_BIG_STAMP") ? atoi(getenv("CVFT_BIG_STAMP")) : 0;
    if (stamp) hipLaunchKernelGGL(gemm_big_kernel<true>, dim3((unsigned)tiles), dim3(512), sm, st, q);
    else hipLaunchKernelGGL(gemm_big_kernel<false>, dim3((unsigned)tiles), dim3(512), sm, st, q);
    cvft_set_kernel_label("gemm_big_kernel<bf16,256,256,2x4,stagger>");
    CVFT_LAUNCH_CHECK("cvft_gemm");
    return 0;
}

extern "C" int cvft_debug_big_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(cvft_big_stamps), sizeof(unsigned long long) * 2 * 16 * 8);
}
