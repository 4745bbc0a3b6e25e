// gemm_p256.hip -- bf16 GEMM for the LLM-sized launches (M ~ 5 000 rows, N = 1024 / 3072 / 4096, K = 1024 .. 4096):
// one 256 x 256 output tile per workgroup, 8 waves as 2 (M) x 4 (N) with 128 x 64 wave tiles, k-tiles of 64, operands
// HBM/L2 -> LDS by LDS-DMA in 16 KB HALF-TILES (128 rows x 128 B) through a ring of ten slots (all 160 KB of the CU),
// the k-loop cut into four phases per k-tile (cdna_hip_programming.md section 5, "The 256^2 8-phase template"):
//
//   phase p of k-tile t:  fragment reads of this phase's quadrant  |  DMA of half-tile 4t + p + 6 (two pieces per wave)
//                         [p == 4: s_waitcnt vmcnt(6) -- k-tile t+1 has landed, three half-tiles stay in flight]
//                         s_barrier ; 16 x v_mfma_f32_16x16x32_bf16 (64 x 32 quadrant x K = 64) ; s_barrier
//
// Waves 4-7 (the lower 128 rows) run one barrier behind waves 0-3, so on every SIMD one wave is in its MFMA section
// while its partner reads fragments and issues DMA pieces.  Per k-tile a wave reads its A half twice 64 rows (phases 1, 3)
// and its W half twice 32 columns (phases 1, 2); phase 4 re-uses registers.  Hazards (global phase g = 4t + p):
//   RAW  a half-tile is read from the phase AFTER the vmcnt + barrier that retired it (waited in phase 4 of k-tile t-1);
//   WAR  the slot of half-tile h is restaged by half-tile h + 10 in phase h + 4: W-top of tile t (last read in phase 2)
//        in phase 4 of the same tile, everything else three or four phases after its last read -- never less than two,
//        which is what the one-barrier lag between the wave groups needs.
// C[M,N] = epilogue( alpha * ( A[M,K] . W[N,K]^T + U[M,R] . Bl[N,R]^T ) ), the epilogue chain / masked rank extension /
// column map are gemm_glds.hip's (gemm_common.h).  SK: split-K -- a workgroup accumulates a k-range, leaves its fp32 tile in a
// workspace slab in register order, and the LAST arriver of a tile (agent-scope ticket, cdna_hip_programming.md section 5,
// "Projection GEMM at M = 256" item 2) sums the slabs and runs the extension + epilogue.
//
// Replaces (reference): lora.py:64-76 for the LLM's linear_q/k/v/out (attention.py:53-80) and w_1 / w_2
// (positionwise_feed_forward.py:47-55), and their dgrads.
#include <stdlib.h>
#include <map>
#include <mutex>
#include "gemm_common.h"

#ifndef P256_DBG
#define P256_DBG 0          // diagnostic builds only (tools/build_p256_dbg.sh): 1 no DMA in the k-loop, 2 fragment reads in the first k-tile only,
#endif                      // 3 no MFMAs (reads kept alive), 4 no barriers in the k-loop -- results are garbage, the timing is the point

struct P256X {
    int tiles_m, tiles_n;       // output tiles
    int S;                      // k-splits per tile (1 = none)
    int nk_total;               // K / 64
    int hb;                     // band height of the tile order (rows of tiles walked column-major inside a band)
    float* ws;                  // SK: [tiles][S][65536] fp32 slabs (or the same count of bf16 when slab_bf16)
    int slab_bf16;              // SK: partial tiles travel as bf16 (half the slab bytes); EVERY partial, the last arriver's own
                                // included, is rounded the same way before the fp32 sum, so the result does not depend on who arrives last
    unsigned* cnt;              // SK: [tiles] arrival tickets (zero between launches)
};

// NB staged row groups (8 rows apart) of one wave's 64 x 64 fp32 tile through the epilogue chain, 8 consecutive columns per lane:
// every global load of the batch is issued before the first use (one memory round trip per batch, not per row), the bias
// comes in registers.  Identity row geometry, no length mask (gemm_p256_launch checks); m = first output row of this lane,
// lr = its row in the staging tile, n = first output column.  Chain order = gemm_epilogue_direct8.
template <int NB>
__device__ __forceinline__ void p256_epilogue_rows(const GP<bf16_t>& p, const float* wl, const float (&bias8)[8], int m, int lr, int cc, int n) {
    float v[NB][8];
    bf16x8 d[NB], r[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const size_t mc = (size_t)min(m + 8 * b, p.M - 1);
        if (p.dact_src) d[b] = *reinterpret_cast<const bf16x8*>(&p.dact_src[mc * p.ldd + n]);
        if (p.residual) r[b] = *reinterpret_cast<const bf16x8*>(&p.residual[mc * p.ldr + n]);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(wl + (lr + 8 * b) * 68 + cc);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(wl + (lr + 8 * b) * 68 + cc + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[b][e] = v0[e] * p.alpha + bias8[e]; v[b][4 + e] = v1[e] * p.alpha + bias8[4 + e]; }
    }
    if (p.preact) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            bf16x8 t;
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = (bf16_t)v[b][e];
            if (m + 8 * b < p.M) *reinterpret_cast<bf16x8*>(&p.preact[(size_t)(m + 8 * b) * p.ldp + n]) = t;
        }
    }
    if (p.act != CVFT_ACT_NONE) {
#pragma unroll
        for (int b = 0; b < NB; ++b) act_apply_vec<8>(p.act, v[b]);
    }
    if (p.dact_src) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float ds[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) ds[e] = (float)d[b][e];
            act_grad_mul_vec<8>(p.dact, v[b], ds);
        }
    }
    if (p.odrop_p > 0.f) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const unsigned long long idx = (unsigned long long)(min(m + 8 * b, p.M - 1) + p.row_off) * (unsigned long long)p.N + n;
            gemm_odrop4(p.odrop_p, p.xdrop_seed, p.odrop_site, idx, v[b]);
            gemm_odrop4(p.odrop_p, p.xdrop_seed, p.odrop_site, idx + 4, v[b] + 4);
        }
    }
    if (p.residual) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[b][e] += (float)r[b][e];
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[b][e];
        if (m + 8 * b < p.M) *reinterpret_cast<bf16x8*>(&p.C[(size_t)(m + 8 * b) * p.ldc + n]) = o;
    }
}

__device__ unsigned long long cvft_p256_stamps[512 * 8];      // diagnostics (CVFT_P256_STAMP=1, tools/bench_p256.py stamps)

// One work item (output tile, or a tile's k-split) of the launch; `item` of `nitems`.
template <bool DX, bool SK, bool STAMP>
__device__ __forceinline__ void p256_item(const GP<bf16_t>& p, const P256X& x, const int item, const int nitems) {
    typedef bf16_t T;
    constexpr int HALF = 16384, RING = 10;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int kg = lane >> 4, l15 = lane & 15;
    const bool rec = STAMP && item < 512 && tid == 0;
    unsigned long long* sb = cvft_p256_stamps + (item & 511) * 8;
    if (rec) sb[0] = __builtin_amdgcn_s_memrealtime();

    // ---- workgroup -> (tile, k-split).  XCD-aware: workgroups b and b + 8 share an XCD, so XCD x takes a contiguous range
    // of the tile order; the order walks bands of hb tile rows column-major, so a range of ~32 tiles is a near-square patch
    // (its A row panels and W column panels stay in that XCD's L2 while the patch streams through k in lock-step).
    int bid = item;
    {
        const int nwg = nitems, q = nwg >> 3, rm = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + loc;
    }
    const int tile = SK ? bid / x.S : bid;
    const int zs = SK ? bid - tile * x.S : 0;
    int tm, tn;
    {
        const int band_sz = x.hb * x.tiles_n;
        const int band = tile / band_sz, rem = tile - band * band_sz;
        const int rows_here = min(x.hb, x.tiles_m - band * x.hb);
        tn = rem / rows_here;
        tm = band * x.hb + (rem - tn * rows_here);
    }
    const int m0 = tm * 256, n0 = tn * 256;
    int kt0 = 0, nk = x.nk_total;
    if (SK) {               // k-tiles [kt0, kt0 + nk): the first (nk_total % S) splits take one more
        const int base = x.nk_total / x.S, extra = x.nk_total - base * x.S;
        kt0 = zs * base + min(zs, extra);
        nk = base + (zs < extra ? 1 : 0);
    }

    // ---- DMA sources.  Half-tile kinds j: 0 = W rows 0..127 of the tile, 1 = W rows 128..255, 2 = A rows 0..127, 3 = A rows
    // 128..255.  A wave moves pieces wid and wid + 8 of each (8 rows x 128 B); lane -> (row piece*8 + lane/8, slot lane%8), the
    // chunk XOR-ed on the SOURCE side so the lane-linear LDS image comes out swizzled (gemm_common.h: A (r>>1)&7, W glds_wswz).
    const char* src[4][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wid + 8 * i) * 8 + (lane >> 3);
        const int gcw = (lane & 7) ^ glds_wswz(r);
        const int gca = (lane & 7) ^ ((r >> 1) & 7);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            src[hh][i] = reinterpret_cast<const char*>(p.W + (size_t)(n0 + 128 * hh + r) * p.ldw + gcw * 8) + (size_t)kt0 * 128;
            const int m = min(m0 + 128 * hh + r, p.M - 1);
            src[2 + hh][i] = reinterpret_cast<const char*>(p.A + (size_t)m * p.lda + gca * 8) + (size_t)kt0 * 128;
        }
    }
    // Plain rank-R extension (U . Bl^T, R <= 64): ONE more k-tile of the same pipeline whose half-tiles come from Bl / U rows
    // (R * 2 bytes each; chunks at or beyond R re-read chunk 0 -- finite numbers -- and the U fragments are masked in
    // registers).  The masked form (DX: one mask per output element and rank tile) cannot be folded, it runs after the loop.
    const bool ext = !DX && p.R > 0 && (!SK || zs == x.S - 1);        // workgroup-uniform
    const int nkl = nk + (ext ? 1 : 0);        // k-tiles of the loop
    const int nh = 4 * nkl;                    // half-tiles of this workgroup's stream
    int hnext = 0;                             // next half-tile to stage
    int sl_stage = 0;                          // its slot
    bool in_loop = false;
    auto stage = [&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        if (hnext < nh && !(P256_DBG == 1 && in_loop)) {                      // wave-uniform
            unsigned char* base = smem + sl_stage * HALF;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_global_load_lds((glb_void_t*)src[j][i], (lds_void_t*)(base + (wid + 8 * i) * 1024), 16, 0, 0);
            if (ext && (hnext >> 2) + 1 == nk) {                    // this kind's next half-tile belongs to the extension tile
                const int rc8 = p.R >> 3;
                int ln = lane;
                asm volatile("" : "+v"(ln));                     // opaque: keeps these addresses from being pre-computed into 16 loop-long registers
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int r = (wid + 8 * i) * 8 + (ln >> 3);
                    if (j < 2) {
                        const int gc = (ln & 7) ^ glds_wswz(r);
                        src[j][i] = reinterpret_cast<const char*>(p.Bl + (size_t)(n0 + 128 * j + r) * p.ldbl + (gc < rc8 ? gc : 0) * 8);
                    } else {
                        const int gc = (ln & 7) ^ ((r >> 1) & 7);
                        const int m = min(m0 + 128 * (j - 2) + r, p.M - 1);
                        src[j][i] = reinterpret_cast<const char*>(p.U + (size_t)m * p.ldu + (gc < rc8 ? gc : 0) * 8);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) src[j][i] += 128;
            }
        }
        ++hnext;
        sl_stage = (sl_stage + 1 == RING) ? 0 : sl_stage + 1;
    };

    // ---- fragment read offsets inside a half-tile (identical to gemm_glds.hip: A rows l15 of a 16-row group, chunk
    // (ks*4 + kg) ^ (l15 >> 1); W rows by the column map, lane-constant XOR)
    const int fx = l15 >> 1;
    const int rd0 = l15 * 128 + ((kg ^ fx) << 4);
    const int rd1 = l15 * 128 + (((4 + kg) ^ fx) << 4);
    const int wrow = 8 * (l15 >> 2) + (l15 & 3), fw = ((l15 >> 1) & 1) | ((l15 >> 2) << 1);
    const int rw0 = ((wc & 1) * 64 + wrow) * 128 + ((kg ^ fw) << 4);
    const int rw1 = ((wc & 1) * 64 + wrow) * 128 + (((4 + kg) ^ fw) << 4);

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: k-tile 0 and three half-tiles of k-tile 1
    static_for<4>([&](auto j) { stage(j); });
    static_for<3>([&](auto j) { stage(j); });
    if (nkl >= 2) wait_vmcnt<6>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (rec) sb[1] = __builtin_amdgcn_s_memrealtime();
    if (wr == 1) __builtin_amdgcn_s_barrier();          // the lower wave group runs one barrier behind

    bf16x8 a[2][4], b[2][4];
    auto mask_a = [&]() __attribute__((always_inline)) {      // extension tile: zero the k >= R part of the U fragments
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const unsigned km = (ks * 32 + kg * 8) < p.R ? 0xffffffffu : 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint4 v = *reinterpret_cast<uint4*>(&a[ks][i]);
                v.x &= km; v.y &= km; v.z &= km; v.w &= km;
                a[ks][i] = *reinterpret_cast<bf16x8*>(&v);
            }
        }
    };
    int s0 = 0;                                         // slot of this k-tile's half-tile 0
    auto k_tile = [&](const int t, auto extc) __attribute__((always_inline)) {
        constexpr bool EXT = decltype(extc)::value;         // the extension tile: U fragments masked in registers
        int sa = s0 + 2 + wr, sb = s0 + (wc >> 1);
        sa = sa >= RING ? sa - RING : sa;
        sb = sb >= RING ? sb - RING : sb;
        const unsigned char* Ab = smem + sa * HALF;
        const unsigned char* Wb = smem + sb * HALF;
        auto read_w = [&](auto j0c) __attribute__((always_inline)) {        // W columns 32 (j0/2) .. +31 of the wave's 64 -> b[.][j0], b[.][j0+1]
            constexpr int j0 = decltype(j0c)::value;
            if (P256_DBG == 2 && t > 0) return;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = j0; j < j0 + 2; ++j) b[ks][j] = *reinterpret_cast<const bf16x8*>(Wb + glds_col(j, 0) * 128 + (ks ? rw1 : rw0));
        };
        auto read_a = [&](auto i0c) __attribute__((always_inline)) {        // A rows 16 i0 .. +63 of the wave's 128 -> a[.][0..3]
            constexpr int i0 = decltype(i0c)::value;
            if (P256_DBG == 2 && t > 0) return;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i) a[ks][i] = *reinterpret_cast<const bf16x8*>(Ab + (i0 + i) * 2048 + (ks ? rd1 : rd0));
            if constexpr (EXT) mask_a();
        };
        auto mma = [&](auto i0c, auto j0c, auto sjc) __attribute__((always_inline)) {  // 64 x 32 quadrant x K = 64: 16 MFMAs between two barriers
            constexpr int i0 = decltype(i0c)::value, j0 = decltype(j0c)::value;
            if (P256_DBG != 4) __builtin_amdgcn_s_barrier();
            if (P256_DBG != 5 && P256_DBG != 7) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int j = j0; j < j0 + 2; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (P256_DBG == 3) asm volatile("" ::"v"(a[ks][i]), "v"(b[ks][j]));
                        else Mma<T>::mma(acc[i0 + i][j], b[ks][j], a[ks][i]);
                    }
                if (P256_DBG == 6 && ks == 0) stage(sjc);          // DMA pieces in the middle of the MFMA cluster
            }
            if (P256_DBG != 5 && P256_DBG != 7) __builtin_amdgcn_s_setprio(0);
            if (P256_DBG != 4) __builtin_amdgcn_s_barrier();
        };
        typedef std::integral_constant<int, 0> I0; typedef std::integral_constant<int, 1> I1;
        typedef std::integral_constant<int, 2> I2; typedef std::integral_constant<int, 3> I3; typedef std::integral_constant<int, 4> I4;
        // ---------------- phase 1: W columns 0..31 + A rows 0..63 -> acc[0..3][0..1]
        if (P256_DBG == 7) __builtin_amdgcn_s_setprio(2);
        read_w(I0{});
        read_a(I0{});
        if (P256_DBG != 6) stage(I3{});
        if (P256_DBG == 7) __builtin_amdgcn_s_setprio(0);
        mma(I0{}, I0{}, I3{});
        // ---------------- phase 2: W columns 32..63 -> acc[0..3][2..3]
        if (P256_DBG == 7) __builtin_amdgcn_s_setprio(2);
        read_w(I2{});
        if (P256_DBG != 6) stage(I0{});
        if (P256_DBG == 7) __builtin_amdgcn_s_setprio(0);
        mma(I0{}, I2{}, I0{});
        // ---------------- phase 3: A rows 64..127 -> acc[4..7][2..3]
        if (P256_DBG == 7) __builtin_amdgcn_s_setprio(2);
        read_a(I4{});
        if (P256_DBG != 6) stage(I1{});
        if (P256_DBG == 7) __builtin_amdgcn_s_setprio(0);
        mma(I4{}, I2{}, I1{});
        // ---------------- phase 4: registers only -> acc[4..7][0..1]; k-tile t+1 must have landed behind this barrier
        if (P256_DBG == 7) __builtin_amdgcn_s_setprio(2);
        if (P256_DBG != 6) stage(I2{});
        if (t + 2 < nkl) wait_vmcnt<(P256_DBG == 6 ? 4 : 6)>();
        else wait_vmcnt<0>();
        if (P256_DBG == 7) __builtin_amdgcn_s_setprio(0);
        mma(I4{}, I0{}, I2{});
        s0 = s0 + 4 >= RING ? s0 + 4 - RING : s0 + 4;
    };
    in_loop = true;
    for (int t = 0; t < nk; ++t) k_tile(t, std::false_type{});
    if (ext) k_tile(nk, std::true_type{});                  // its own copy of the body: a mask branch inside the loop made hipcc
                                                            // read every A fragment into temporaries and wait for them ahead of the barrier
    if (wr == 0) __builtin_amdgcn_s_barrier();          // balance the lag
    if (rec) sb[2] = __builtin_amdgcn_s_memrealtime();

    // ---- split-K: leave the tile in the workspace in register order; the last arriver carries on with the sum
    if constexpr (SK) {
        float* slab = x.ws + ((size_t)tile * x.S + zs) * 65536;
        bf16_t* slabh = reinterpret_cast<bf16_t*>(x.ws) + ((size_t)tile * x.S + zs) * 65536;
        unsigned* flag = reinterpret_cast<unsigned*>(smem);             // ring is dead (every DMA waited for, all reads done)
        if (x.slab_bf16) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    bf16x8 h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { h[e] = (bf16_t)acc[i][2 * jp][e]; h[4 + e] = (bf16_t)acc[i][2 * jp + 1][e]; }
                    *reinterpret_cast<bf16x8*>(slabh + ((i * 2 + jp) * 512 + tid) * 8) = h;
                    // the own partial enters the sum rounded like everybody else's
#pragma unroll
                    for (int e = 0; e < 4; ++e) { acc[i][2 * jp][e] = (float)h[e]; acc[i][2 * jp + 1][e] = (float)h[4 + e]; }
                }
        } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<float4*>(slab + ((i * 4 + j) * 512 + tid) * 4) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned prev = __hip_atomic_fetch_add(&x.cnt[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned last = (prev == (unsigned)(x.S - 1)) ? 1u : 0u;
            if (last) {
                __hip_atomic_store(&x.cnt[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            *flag = last;
        }
        __syncthreads();
        if (*flag == 0u) return;
        // fixed summation order (slab 0 + slab 1 + ...) whoever arrives last, the own tile taken from registers: results do
        // not depend on the arrival order
        if (x.slab_bf16) {
            const bf16_t* base = reinterpret_cast<const bf16_t*>(x.ws) + ((size_t)tile * x.S) * 65536;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    f32x4 f0 = {0.f, 0.f, 0.f, 0.f}, f1 = {0.f, 0.f, 0.f, 0.f};
                    for (int z = 0; z < x.S; ++z) {
                        if (z == zs) {
                            f0 += acc[i][2 * jp];
                            f1 += acc[i][2 * jp + 1];
                        } else {
                            const bf16x8 h = *reinterpret_cast<const bf16x8*>(base + (size_t)z * 65536 + ((i * 2 + jp) * 512 + tid) * 8);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { f0[e] += (float)h[e]; f1[e] += (float)h[4 + e]; }
                        }
                    }
                    acc[i][2 * jp] = f0;
                    acc[i][2 * jp + 1] = f1;
                }
        } else
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 f = acc[i][j];
                if (zs != 0) {
                    const float4 v = *reinterpret_cast<const float4*>(x.ws + ((size_t)tile * x.S) * 65536 + ((i * 4 + j) * 512 + tid) * 4);
                    f = f32x4{v.x, v.y, v.z, v.w};
                }
                for (int z = 1; z < x.S; ++z) {
                    if (z == zs) {
                        f += acc[i][j];
                    } else {
                        const float4 v = *reinterpret_cast<const float4*>(x.ws + ((size_t)tile * x.S + z) * 65536 + ((i * 4 + j) * 512 + tid) * 4);
                        f += f32x4{v.x, v.y, v.z, v.w};
                    }
                }
                acc[i][j] = f;
            }
    }

    // ---- masked rank-R extension (lora_dropout dgrad; R <= 64), one 32-wide rank step at a time: fragment-shaped direct loads
    // (rows of U / Bl are R * 2 bytes: L2-resident), acc += mask_t / (1 - p) * (U_t . Bl_t^T) per 16-wide rank tile t with the mask
    // over the OUTPUT elements -- a lane owns 4 consecutive n of one row, exactly one keep4 group (as gemm_glds.hip)
    const int mrow0 = m0 + wr * 128, ncol0 = n0 + wc * 64;
    if constexpr (DX) {
        const int nrs = (p.R + 31) >> 5;
        const unsigned thr = cvft_drop_thr(p.xdrop_p);
        const float inv = 1.f / (1.f - p.xdrop_p);
        for (int s = 0; s < nrs; ++s) {
            const int kk = s * 32 + kg * 8;
            const int kkc = kk < p.R ? kk : 0;
            const unsigned keep = kk < p.R ? 0xffffffffu : 0u;
            uint4 ua[8], ub[4];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = min(mrow0 + i * 16 + l15, p.M - 1);
                ua[i] = *reinterpret_cast<const uint4*>(p.U + (size_t)m * p.ldu + kkc);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = ncol0 + glds_col(j, l15);
                ub[j] = *reinterpret_cast<const uint4*>(p.Bl + (size_t)n * p.ldbl + kkc);
            }
#pragma unroll
            for (int th = 0; th < 2; ++th) {
                const int tr = 2 * s + th;                                   // 16-wide rank tile
                if (tr * 16 < p.R) {                                         // wave-uniform
                    const unsigned km = ((kg >> 1) == th) ? keep : 0u;       // this tile's 16 k of the 32-wide step
                    const unsigned long long key = cvft_drop_key(p.xdrop_seed, p.xdrop_sites[tr]);
                    bf16x8 fb[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<bf16x8*>(&ub[j]);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        uint4 v = ua[i];
                        v.x &= km; v.y &= km; v.z &= km; v.w &= km;
                        const bf16x8 fa = *reinterpret_cast<bf16x8*>(&v);
                        const unsigned long long m = (unsigned long long)(min(mrow0 + i * 16 + l15, p.M - 1) + p.row_off);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            f32x4 a2 = f32x4{0.f, 0.f, 0.f, 0.f};
                            Mma<T>::mma(a2, fb[j], fa);
                            const int n = ncol0 + glds_col(j, 4 * kg);
                            bool k4[4];
                            cvft_keep4(key, (m * (unsigned long long)p.N + n) >> 2, thr, k4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[i][j][e] += k4[e] ? a2[e] * inv : 0.f;
                        }
                    }
                }
            }
        }
    }

    // ---- epilogue.  The accumulators go through a wave-private fp32 staging tile in the (dead) ring -- 64 rows x 64 columns at a
    // time, rows 272 B apart -- and come back row-major: a lane takes 8 consecutive columns, 8 lanes one 128-byte output line, a
    // wave instruction 8 whole lines.  The bias / act / act' / dropout / residual chain then runs in a ROLLED loop: one copy of
    // gemm_epilogue_direct8 in the instruction stream instead of sixteen (unrolled over the accumulator registers the epilogue
    // was ~33 000 instructions, every copy fetched cold: ~2/3 of the launch's fixed cost at K = 1024).  No barrier: nobody
    // reads the ring after the loop's last barrier (every wave's fragment reads and DMA have completed behind it).
    if (rec) sb[3] = __builtin_amdgcn_s_memrealtime();
    float* wl = reinterpret_cast<float*>(smem) + wid * (64 * 68);
    const int rr = lane >> 3, cc = (lane & 7) * 8;
    const int n = ncol0 + cc;
    float bias8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
    if (p.bias) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
        bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w; bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<f32x4*>(wl + (i * 16 + l15) * 68 + glds_col(j, 4 * kg)) = acc[4 * hf + i][j];
#pragma unroll 1
        for (int it = 0; it < 2; ++it) p256_epilogue_rows<4>(p, wl, bias8, mrow0 + hf * 64 + it * 32 + rr, it * 32 + rr, cc, n);
    }
    if (rec) {
        sb[4] = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sb[5] = __builtin_amdgcn_s_memrealtime();
        sb[6] = (unsigned long long)tile;
    }
}

// The launch: gridDim.x workgroups walk the items gridDim.x apart.  gridDim.x == items is one item per workgroup (the plain
// launch); a smaller grid (a multiple of 8, so that an item's index modulo 8 -- the XCD its neighbours share -- stays its
// workgroup's) leaves CUs to the other chains of the step at the price of more rounds (CVFT_P256_GRID).
template <bool DX, bool SK, bool STAMP = false>
__global__ void __launch_bounds__(512, 2) gemm_p256_kernel(GP<bf16_t> p, P256X x, int nitems) {
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        p256_item<DX, SK, STAMP>(p, x, item, nitems);
        if (item + (int)gridDim.x < nitems) __syncthreads();      // the next item's DMA re-uses the ring the epilogue staged through
    }
}

// ---------------------------------------------------------------- host side
namespace {
// Split-K workspaces: a pool of equal-sized slots, one per launch stream (chains on different streams run concurrently; a
// captured step launches from the capture stream, which need not be the stream of the eager warm-up runs).  Slots are handed
// out on first sight of a stream without allocating; the pool grows (all slots at once, old buffers kept: captured steps hold
// their addresses) only outside stream capture -- hipMalloc is illegal inside one -- so the eager runs that precede every
// capture size it.
struct P256Ws { float* ws = nullptr; unsigned* cnt = nullptr; };
constexpr int P256_SLOTS = 8;
constexpr size_t P256_NCNT = 4096;
std::mutex g_ws_mu;
P256Ws g_pool[P256_SLOTS];
size_t g_pool_floats = 0;
std::map<hipStream_t, int> g_slot_of;

int p256_workspace(hipStream_t st, size_t floats, size_t tiles, P256Ws* out) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    if (tiles > P256_NCNT) { cvft_set_error("cvft_gemm: split-K over %zu tiles (counter table holds %zu)", tiles, P256_NCNT); return -1; }
    if (g_pool_floats < floats) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
            cvft_set_error("cvft_gemm: split-K workspace must grow to %zu bytes per stream inside a stream capture: run the step eagerly once first", floats * 4);
            return -2;
        }
        const size_t want = floats + floats / 4;                       // head room: a slightly larger batch layout re-uses it
        for (int i = 0; i < P256_SLOTS; ++i) {
            float* nw = nullptr; unsigned* nc = nullptr;
            if (hipMalloc(&nw, want * sizeof(float)) != hipSuccess || hipMalloc(&nc, P256_NCNT * sizeof(unsigned)) != hipSuccess ||
                hipMemset(nc, 0, P256_NCNT * sizeof(unsigned)) != hipSuccess) {
                cvft_set_error("cvft_gemm: split-K workspace of %zu bytes: hipMalloc failed", want * 4);
                return -2;
            }
            g_pool[i].ws = nw; g_pool[i].cnt = nc;
        }
        if (hipDeviceSynchronize() != hipSuccess) { cvft_set_error("cvft_gemm: split-K workspace: device synchronize failed"); return -2; }
        g_pool_floats = want;
    }
    auto it = g_slot_of.find(st);
    if (it == g_slot_of.end()) {
        if ((int)g_slot_of.size() >= P256_SLOTS) { cvft_set_error("cvft_gemm: split-K launches from more than %d streams", P256_SLOTS); return -1; }
        it = g_slot_of.emplace(st, (int)g_slot_of.size()).first;
    }
    *out = g_pool[it->second];
    return 0;
}

template <bool DX, bool SK, bool STAMP = false>
int p256_launch_t(const GP<bf16_t>& p, const P256X& x, hipStream_t st) {
    auto kern = gemm_p256_kernel<DX, SK, STAMP>;
    static bool attr_set = false;
    if (!attr_set) {
        attr_set = true;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        if (e != hipSuccess) { cvft_set_error("cvft_gemm: hipFuncSetAttribute(163840) failed: %s", hipGetErrorString(e)); return -2; }
    }
    // Workgroups resident at once.  CVFT_P256_GRID: 0 = one per item, n = at most n; unset = auto: while three or more chains
    // share the chip (cvft_set_concurrent_chains: the joint step) HALF the CUs, two items per workgroup.  This kernel owns every
    // CU it runs on (160 KB of LDS); a 252-item launch on 252 CUs stops the step's two Flow chains -- ~1 000 short, dependent
    // launches each, whose latency the step follows -- for its whole duration.  Measured in the step, same box, 40 steps
    // (DESIGN section 14): joint 21.30 / 21.30 ms with one workgroup per item, 20.59 / 20.60 with 128, 21.03 with 88 (three
    // rounds), although the launch itself goes from 45 to 65 us; alone on the chip (llm_only, one chain) the full grid stays.
    static const int grid_set = getenv("CVFT_P256_GRID") ? atoi(getenv("CVFT_P256_GRID")) : -1;
    const int grid_env = grid_set >= 0 ? grid_set : (cvft_concurrent_chains() >= 3 ? 128 : 0);
    const int nitems = x.tiles_m * x.tiles_n * x.S;
    int grid = nitems;
    if (grid_env >= 8 && grid_env < nitems && !SK) {     // (split-K: the last arriver of a tile must not queue behind its own partners)
        // rounds first (as few as the cap allows), then the smallest grid that still needs no more rounds: equal work per workgroup
        const int cap = grid_env & ~7, rounds = (nitems + cap - 1) / cap;
        grid = (((nitems + rounds - 1) / rounds) + 7) & ~7;
        if (grid > cap) grid = cap;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), 163840, st, p, x, nitems);
    cvft_set_kernel_label("gemm_p256_kernel<bf16,256,256,2,4,ring10>%s%s", SK ? ",splitk" : "", DX ? ",xdrop" : "");
    CVFT_LAUNCH_CHECK("cvft_gemm");
    return 0;
}
}  // namespace

// Returns 1 when the launch is not eligible (the caller carries on with gemm_glds.hip's tiles).
int gemm_p256_launch(const GP<bf16_t>& p_in, hipStream_t st) {
    static const int mode = getenv("CVFT_P256") ? atoi(getenv("CVFT_P256")) : 1;        // 0 off, 1 auto, 2 whenever legal
    static const int split_env = getenv("CVFT_P256_SPLIT") ? atoi(getenv("CVFT_P256_SPLIT")) : 0;   // 0 auto, >= 1 forced
    static const int hb_env = getenv("CVFT_P256_HB") ? atoi(getenv("CVFT_P256_HB")) : 0;
    static const int rowsplit_env = getenv("CVFT_P256_ROWSPLIT") ? atoi(getenv("CVFT_P256_ROWSPLIT")) : 1;
    static const int stamp_env = getenv("CVFT_P256_STAMP") ? atoi(getenv("CVFT_P256_STAMP")) : 0;
    static const int slabh_env = getenv("CVFT_P256_SLAB_BF16") ? atoi(getenv("CVFT_P256_SLAB_BF16")) : 1;
    if (mode == 0) return 1;
    GP<bf16_t> p = p_in;
    p.direct_epi = 1;
    const bool ident = p.ntaps == 1 && p.tap_off[0] == 0 && p.in_stride == 1 && p.Tin == p.Tm && !p.in_len &&
                       p.Tm == p.M && p.out_stride == 1 && p.out_off == 0 && !p.out_len;
    if (!ident || p.fuse || p.K % 64 != 0 || p.N % 256 != 0 || !p.vecA || !p.vecW) return 1;
    if (!glds_direct_epilogue(p) || !glds_wide_epilogue(p)) return 1;
    if (p.R > 0 && (p.R % 8 != 0 || p.R > 64 || !p.vecU || !p.vecB)) return 1;
    if (p.xdrop_p > 0.f && (p.R % 16 != 0 || p.R <= 0)) return 1;
    P256X x;
    x.tiles_m = (p.M + 255) / 256; x.tiles_n = p.N / 256; x.nk_total = p.K / 64; x.S = 1; x.ws = nullptr; x.cnt = nullptr;
    x.slab_bf16 = slabh_env ? 1 : 0;
    long tiles = (long)x.tiles_m * x.tiles_n;
    // k-splits (CVFT_P256_SPLIT=n, or -1 = as many as fill the 256 CUs with >= 12 k-tiles each): opt-in.  Measured at
    // 5328 x 1024 x 4096 / x 3072 (84 tiles, three splits): 74 / 64 us against 61 / 50 us for one round of 96x256 tiles -- every
    // split pays the prologue and 256 KB of fp32 slab out and (the last arriver) 512 KB back, more than the idle CUs cost.
    int S = 1;
    if (split_env >= 1) S = split_env;
    else if (split_env < 0 && tiles <= 128) { S = (int)(256 / tiles); while (S > 1 && x.nk_total / S < 12) --S; }
    if (S > x.nk_total) S = x.nk_total;
    if (mode == 1) {
        // auto: LLM-sized launches only -- at least ~3/4 of a round of tiles (after splitting) and a k-loop that amortises the
        // prologue / epilogue; everything else is faster on two co-resident 128x128 blocks per CU or on the 64x64 tiles
        if (p.M < 2048 || x.nk_total < 8) return 1;
        static const int narrow_env = getenv("CVFT_P256_NARROW") ? atoi(getenv("CVFT_P256_NARROW")) : 0;
        // (CVFT_P256_NARROW=n: also launches of n .. 191 tiles, one per CU on a third of the chip -- the N = 1024 shapes at M ~ 5 000)
        if (tiles * S < 192 && !(narrow_env > 0 && tiles * S >= narrow_env)) return 1;
    }
    // Row split: a grid between one and ~1.6 rounds of the 256 CUs (N = 4096 at M = 5328: 336 tiles) would hold the chip for two
    // rounds.  The first 256 / tiles_n tile rows make exactly <= one round here; the rows behind them go to gemm_glds.hip's
    // 128x128 tiles (two blocks per CU) as a second launch whose first blocks start under this launch's tail.  The dropout masks
    // index the whole tensor: row_off.
    if (rowsplit_env && S == 1 && tiles > 256 && tiles <= 416 && x.tiles_n <= 256) {
        const int rows_m = 256 / x.tiles_n;
        if (rows_m >= 1 && rows_m < x.tiles_m) {
            const int M1 = rows_m * 256;
            GP<bf16_t> q = p_in;
            q.M = p.M - M1; q.Tm = q.M; q.Tin = q.M; q.Tout = q.M;
            q.A = p.A + (size_t)M1 * p.lda;
            if (p.U) q.U = p.U + (size_t)M1 * p.ldu;
            if (p.preact) q.preact = p.preact + (size_t)M1 * p.ldp;
            if (p.dact_src) q.dact_src = p.dact_src + (size_t)M1 * p.ldd;
            if (p.residual) q.residual = p.residual + (size_t)M1 * p.ldr;
            q.C = p.C + (size_t)M1 * p.ldc;
            q.row_off = p.row_off + M1;
            p.M = M1; p.Tm = M1; p.Tin = M1; p.Tout = M1;
            x.tiles_m = rows_m; tiles = (long)x.tiles_m * x.tiles_n;
            int rc = gemm_glds_launch(q, st, -256);               // cfg -256: do not come back here
            if (rc != 0) { if (rc == 1) cvft_set_error("cvft_gemm: row split: the remainder launch is not eligible for the LDS-DMA kernels"); return rc < 0 ? rc : -1; }
        }
    }
    x.S = S;
    // band height: patches of ~32 tiles (one XCD's CUs) as square as the grid allows
    int hb = hb_env > 0 ? hb_env : 1;
    if (hb_env <= 0) {
        long best = -1;
        for (int h = 1; h <= x.tiles_m && h <= 32; ++h) {
            const long w = (32 + h - 1) / h;                       // columns a 32-tile patch spans
            const long cost = h + (w > x.tiles_n ? x.tiles_n : w);
            if (best < 0 || cost < best) { best = cost; hb = h; }
        }
    }
    x.hb = hb;
    if (S > 1) {
        P256Ws w;
        int rc = p256_workspace(st, (size_t)tiles * S * 65536, (size_t)tiles, &w);
        if (rc) return rc;
        x.ws = w.ws; x.cnt = w.cnt;
        return p.xdrop_p > 0.f ? p256_launch_t<true, true>(p, x, st) : p256_launch_t<false, true>(p, x, st);
    }
    if (stamp_env && p.xdrop_p <= 0.f) return p256_launch_t<false, false, true>(p, x, st);
    return p.xdrop_p > 0.f ? p256_launch_t<true, false>(p, x, st) : p256_launch_t<false, false>(p, x, st);
}

extern "C" int cvft_debug_p256_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(cvft_p256_stamps), sizeof(unsigned long long) * 512 * 8);
}
