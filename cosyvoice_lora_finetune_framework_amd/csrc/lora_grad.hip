// lora_grad.hip -- LoRA adapter gradients on the matrix cores (bf16, deterministic slab mode).
//
//   slab_y[j, c] = sum_{m in rows of slab y} Rk[m, j] * Wd[m, c]          (dA = V^T X,  dB^T = U^T dY)
//
// Both operands are stored with the reduction index m as the SLOW index, i.e. transposed for the MFMA's
// k-contiguous operand layout.  gfx950's ds_read_b64_tr_b16 does that transpose for free on the LDS read
// (cdna_hip_programming.md T10), so the wide operand goes HBM/L2 -> LDS by LDS-DMA in its natural row-major
// form (128-byte row segments, source-side XOR swizzle for the transposed reads) and is never touched by VALU.
// One wavefront owns a 64-column stripe and a contiguous row range: no cross-wave reduction, no barrier, no
// atomics; accumulators go straight to the slab with 64-byte coalesced stores.  The VALU kernel this replaces
// (gemm.hip lora_rank_accum_vec_kernel) spent 128 FMAs + conversions per 16-byte load and a 128 KB LDS reduction
// per 8 KB of input.
//
// Replaces (reference): the autograd of lora.py:71-76 (grad of lora_A / lora_B).
#include "common.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;

__device__ __forceinline__ bf16x8 tr_read8(const unsigned char* lo, const unsigned char* hi) {
    bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)lo);
    bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)hi);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// RB = r / 16.  Block = 4 wavefronts = 4 x 64 columns; blockIdx.y = slab; each wave walks its slab's rows in
// 32-row chunks (MFMA k = 32), double-buffered in a private LDS region.
struct RankProb {             // one gradient product: slabs of  Rk^T Wd  ([r, Cn], or [Cn, r] when transpose_out)
    int Cn; const bf16_t* Wd; int ldw; const bf16_t* Rk; int ldr; float* out; int ldo; int transpose_out;
    size_t part_stride; int rows_per_block;
};
struct RankPair { RankProb p[4]; };      // up to four problems per launch (blockIdx.z)

template <int RB>
__device__ __forceinline__ void lora_rank_body(int M, const RankProb& pr) {
    const int Cn = pr.Cn, ldw = pr.ldw, ldr = pr.ldr, ldo = pr.ldo, transpose_out = pr.transpose_out;
    const int rows_per_block = pr.rows_per_block;
    const bf16_t* __restrict__ Wd = pr.Wd;
    const bf16_t* __restrict__ Rk = pr.Rk;
    float* __restrict__ out = pr.out;
    const size_t part_stride = pr.part_stride;
    constexpr int R = RB * 16;
    constexpr int WD_BYTES = 32 * 128;               // 32 rows x 64 columns
    constexpr int RK_BYTES = 32 * R * 2;
    constexpr int BUF = WD_BYTES + RK_BYTES;
    constexpr int RCH = R / 8;                       // 16-byte chunks per Rk row
    constexpr int RK_INS = 32 * RCH / 64;            // wave-wide 16-byte loads per 32-row chunk (chunk id = i*64 + lane)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // rows_per_block % 128 == 0 ("stacked"): the four waves share ONE 64-column stripe and each walks a quarter of the slab's rows;
    // their four partial tiles meet in LDS and leave as one.  Same number of waves and chunks per wave as four stripes over a
    // quarter of the rows, a quarter of the slabs to write and to reduce (the LLM's adapter products were 42 slabs each,
    // ~0.8 GB of slab traffic per step each way and a 215 us reduce on the step's serial tail).
    const bool stacked = (rows_per_block & 127) == 0;
    const int c0 = stacked ? blockIdx.x * 64 : (blockIdx.x * 4 + wid) * 64;
    if (c0 >= Cn) return;                            // not stacked: whole wave, no barriers; stacked: whole workgroup
    const int mb0 = blockIdx.y * rows_per_block;
    if (mb0 >= M) return;                            // the grid is sized for the problem with more slabs / columns
    const int me0 = min(M, mb0 + rows_per_block);
    const int mb = stacked ? mb0 + wid * (rows_per_block >> 2) : mb0;
    const int me = stacked ? min(me0, mb + (rows_per_block >> 2)) : me0;
    const int nchunk = max(0, (me - mb + 31) >> 5);
    unsigned char* my = smem + wid * 2 * BUF;

    // DMA source of the wide operand: lane -> (row i*8 + lane/8, slot lane%8); slot s of row r holds global chunk
    // s ^ f(r), f(r) = 2*bit1(r) | 4*bit3(r), which makes the transposed reads below bank-conflict free.  Rows past
    // the slab end are clamped to the last valid row (their Rk rows are zero), chunks past Cn to the last valid chunk.
    const int nchk = Cn >> 3;
    int w_row[4], w_chk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = i * 8 + (lane >> 3);
        const int f = (((r >> 1) & 1) << 1) | (((r >> 3) & 1) << 2);
        w_row[i] = r;
        w_chk[i] = min((c0 >> 3) + ((lane & 7) ^ f), nchk - 1);
    }
    auto issue = [&](int chunk, int buf) __attribute__((always_inline)) {
        const int m0 = mb + chunk * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = min(m0 + w_row[i], me - 1);
            const bf16_t* g = Wd + (size_t)m * ldw + w_chk[i] * 8;
            __builtin_amdgcn_global_load_lds((glb_void_t*)g, (lds_void_t*)(my + buf * BUF + i * 1024), 16, 0, 0);
        }
    };
    // narrow operand: guarded register loads (zero rows past the slab end), written to LDS row-major
    uint4 rk[RK_INS];
    auto load_rk = [&](int chunk) __attribute__((always_inline)) {
        const int m0 = mb + chunk * 32;
#pragma unroll
        for (int i = 0; i < RK_INS; ++i) {
            const int id = i * 64 + lane;
            const int m = m0 + id / RCH;
            rk[i] = m < me ? *reinterpret_cast<const uint4*>(Rk + (size_t)m * ldr + (id % RCH) * 8) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_rk = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < RK_INS; ++i)
            *reinterpret_cast<uint4*>(my + buf * BUF + WD_BYTES + (i * 64 + lane) * 16) = rk[i];
    };

    f32x4 acc[RB][4];
#pragma unroll
    for (int jb = 0; jb < RB; ++jb)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[jb][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposed-read addresses: lane 4q+p of 16-lane group kg supplies row 8kg + q (+4), columns 4p..4p+3
    const int kg = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int rlo = 8 * kg + q, rhi = rlo + 4;
    const int flo = (((rlo >> 1) & 1) << 1) | (((rlo >> 3) & 1) << 2);
    const int fhi = (((rhi >> 1) & 1) << 1) | (((rhi >> 3) & 1) << 2);
    int b_lo[4], b_hi[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int ch = nt * 2 + (pp >> 1);
        b_lo[nt] = rlo * 128 + ((ch ^ flo) << 4) + 8 * (pp & 1);
        b_hi[nt] = rhi * 128 + ((ch ^ fhi) << 4) + 8 * (pp & 1);
    }
    const int a_lo = WD_BYTES + rlo * (R * 2) + pp * 8, a_hi = WD_BYTES + rhi * (R * 2) + pp * 8;

    if (nchunk > 0) {                                          // (stacked: a wave whose quarter starts past M only joins the reduction)
        issue(0, 0);
        load_rk(0);
    }
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // chunk c: DMA landed, Rk registers arrived
        store_rk(buf);
        if (c + 1 < nchunk) {                                  // wave-uniform
            issue(c + 1, buf ^ 1);
            load_rk(c + 1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned char* bb = my + buf * BUF;
        bf16x8 a[RB], b[4];
#pragma unroll
        for (int jb = 0; jb < RB; ++jb) a[jb] = tr_read8(bb + a_lo + jb * 32, bb + a_hi + jb * 32);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) b[nt] = tr_read8(bb + b_lo[nt], bb + b_hi[nt]);
#pragma unroll
        for (int jb = 0; jb < RB; ++jb)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[jb][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[jb], b[nt], acc[jb][nt], 0, 0, 0);
    }

    // acc[jb][nt][i] = slab[j = jb*16 + kg*4 + i][c = c0 + nt*16 + (lane & 15)]
    float* base = out + (size_t)blockIdx.y * part_stride;
    const int l15 = lane & 15;
    if (stacked) {
        // partial tiles into the wave's own staging region (RB * 4 KB <= 2 * BUF), then wave w sums and stores column tile nt = w
        f32x4* red = reinterpret_cast<f32x4*>(my);
#pragma unroll
        for (int jb = 0; jb < RB; ++jb)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) red[(jb * 4 + nt) * 64 + lane] = acc[jb][nt];
        __syncthreads();
        const int cl = c0 + wid * 16 + l15;
#pragma unroll
        for (int jb = 0; jb < RB; ++jb) {
            f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) sacc += reinterpret_cast<const f32x4*>(smem + ww * 2 * BUF)[(jb * 4 + wid) * 64 + lane];
            if (cl >= Cn) continue;
            const int j = jb * 16 + kg * 4;
            if (transpose_out) {
                *reinterpret_cast<float4*>(&base[(size_t)cl * ldo + j]) = make_float4(sacc[0], sacc[1], sacc[2], sacc[3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) base[(size_t)(j + i) * ldo + cl] = sacc[i];
            }
        }
        return;
    }
#pragma unroll
    for (int jb = 0; jb < RB; ++jb)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int cl = c0 + nt * 16 + l15;
            if (cl >= Cn) continue;
            const int j = jb * 16 + kg * 4;
            if (transpose_out) {
                *reinterpret_cast<float4*>(&base[(size_t)cl * ldo + j]) =
                    make_float4(acc[jb][nt][0], acc[jb][nt][1], acc[jb][nt][2], acc[jb][nt][3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) base[(size_t)(j + i) * ldo + cl] = acc[jb][nt][i];
            }
        }
}

// blockIdx.z selects the problem: a LoRA layer's dA and dB (same row count M, same rank) go out as ONE launch.
template <int RB>
__global__ void __launch_bounds__(256) lora_rank_mfma_kernel(int M, RankPair prob_pair) {
    lora_rank_body<RB>(M, prob_pair.p[blockIdx.z]);
}

// Batch form: the problems of MANY layers (each with its own row count), one per blockIdx.z, passed by value in the
// kernel arguments (no device-side table: nothing to upload, so the launch is hipGraph-capturable as is).
// A backward pass defers the small adapter-gradient products to its end (nothing downstream reads them) and issues
// them here as a few chip-filling launches instead of one latency-bound launch per layer on the dgrad chain.
struct RankProbM { const bf16_t* Wd; const bf16_t* Rk; float* out; int M, Cn, ldw, ldr, rows_per_block, transpose_out; };
#define CVFT_RANK_BATCH 64
struct RankBatch { RankProbM p[CVFT_RANK_BATCH]; };          // 64 x 48 B = 3 KB of the 4 KB kernel-argument segment
template <int RB>
__global__ void __launch_bounds__(256) lora_rank_mfma_batch_kernel(RankBatch batch) {
    const RankProbM& e = batch.p[blockIdx.z];
    RankProb pr;
    pr.Cn = e.Cn; pr.Wd = e.Wd; pr.ldw = e.ldw; pr.Rk = e.Rk; pr.ldr = e.ldr; pr.out = e.out;
    pr.transpose_out = e.transpose_out; pr.rows_per_block = e.rows_per_block;
    pr.ldo = pr.transpose_out ? RB * 16 : pr.Cn;
    pr.part_stride = (size_t)(RB * 16) * pr.Cn;
    lora_rank_body<RB>(e.M, pr);
}

static bool rank_ok(int Cn, int r, const void* Wd, int ldw, const void* Rk, int ldr, const float* part, int rows_per_block) {
    return (r == 16 || r == 32 || r == 48 || r == 64) && Cn % 8 == 0 && ldw % 8 == 0 && ldr % 8 == 0 &&
           (reinterpret_cast<uintptr_t>(Wd) & 15) == 0 && (reinterpret_cast<uintptr_t>(Rk) & 15) == 0 &&
           rows_per_block > 0 && rows_per_block % 32 == 0 && (reinterpret_cast<uintptr_t>(part) & 15) == 0;
}
static RankProb make_prob(int Cn, int r, const void* Wd, int ldw, const void* Rk, int ldr, float* part, int transpose_out,
                          int rows_per_block) {
    RankProb q;
    q.Cn = Cn; q.Wd = (const bf16_t*)Wd; q.ldw = ldw; q.Rk = (const bf16_t*)Rk; q.ldr = ldr; q.out = part;
    q.ldo = transpose_out ? r : Cn; q.transpose_out = transpose_out; q.part_stride = (size_t)r * Cn; q.rows_per_block = rows_per_block;
    return q;
}
static void rank_launch(int M, int r, const RankPair& pp, int nprob, hipStream_t st) {
    int gx = 0, gy = 0;
    for (int i = 0; i < nprob; ++i) {
        gx = max(gx, pp.p[i].rows_per_block % 128 == 0 ? (pp.p[i].Cn + 63) / 64 : (pp.p[i].Cn + 255) / 256);
        gy = max(gy, (M + pp.p[i].rows_per_block - 1) / pp.p[i].rows_per_block);
    }
    dim3 grid(gx, gy, nprob);
    const size_t sm = (size_t)4 * 2 * (32 * 128 + 32 * r * 2);
#define RM_LAUNCH(RBv)                                                                                                   \
    do {                                                                                                                 \
        auto kern = lora_rank_mfma_kernel<RBv>;                                                                          \
        static bool attr_set = false;                                                                                    \
        if (sm > 48 * 1024 && !attr_set) {                                                                               \
            attr_set = true;                                                                                             \
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); \
        }                                                                                                                \
        hipLaunchKernelGGL(kern, grid, dim3(256), sm, st, M, pp);                                                        \
    } while (0)
    if (r == 16) RM_LAUNCH(1); else if (r == 32) RM_LAUNCH(2); else if (r == 48) RM_LAUNCH(3); else RM_LAUNCH(4);
#undef RM_LAUNCH
}

// Slab mode only (part_stride = r * Cn): returns 1 when the operands are not eligible.
int lora_rank_mfma_launch(int M, int Cn, int r, const void* Wd, int ldw, const void* Rk, int ldr, float* part,
                          int transpose_out, int rows_per_block, hipStream_t st) {
    if (!rank_ok(Cn, r, Wd, ldw, Rk, ldr, part, rows_per_block)) return 1;
    RankPair pp;
    pp.p[0] = make_prob(Cn, r, Wd, ldw, Rk, ldr, part, transpose_out, rows_per_block);
    pp.p[1] = pp.p[2] = pp.p[3] = pp.p[0];
    rank_launch(M, r, pp, 1, st);
    return 0;
}

// dA and dB of one LoRA layer in one launch (bf16, slab mode):
//   partA[s] [r, K]  = V^T X  over row block s of size rpbA;   partB[s] [N, r] = dY^T U  over row block s of size rpbB
extern "C" int cvft_lora_rank_partial_pair(int M, int r, int K, const void* X, int ldx, const void* V, int ldv, float* partA,
                                           int rpbA, int N, const void* dY, int ldy, const void* U, int ldu, float* partB,
                                           int rpbB, void* stream) {
    CVFT_CHECK_ARG(M > 0 && X && V && partA && dY && U && partB, "cvft_lora_rank_partial_pair: bad args");
    CVFT_CHECK_ARG(rank_ok(K, r, X, ldx, V, ldv, partA, rpbA) && rank_ok(N, r, dY, ldy, U, ldu, partB, rpbB),
                   "cvft_lora_rank_partial_pair: bf16 operands must be 16-byte aligned, widths %% 8 == 0, r in {16,32,48,64}, rows_per_block %% 32 == 0");
    RankPair pp;
    pp.p[0] = make_prob(K, r, X, ldx, V, ldv, partA, 0, rpbA);
    pp.p[1] = make_prob(N, r, dY, ldy, U, ldu, partB, 1, rpbB);
    pp.p[2] = pp.p[3] = pp.p[0];
    rank_launch(M, r, pp, 2, (hipStream_t)stream);
    CVFT_LAUNCH_CHECK("cvft_lora_rank_partial_pair");
    return 0;
}

// Up to four slab products of the same row count M and rank r in one launch (the three dA of stacked q|k|v adapters
// under lora_dropout have three different dropped inputs).
extern "C" int cvft_lora_rank_partial_multi(int M, int r, int n, const cvft_rank_prob* probs, void* stream) {
    CVFT_CHECK_ARG(M > 0 && n >= 1 && n <= 4 && probs, "cvft_lora_rank_partial_multi: 1 <= n <= 4");
    RankPair pp;
    for (int i = 0; i < n; ++i) {
        const cvft_rank_prob& q = probs[i];
        CVFT_CHECK_ARG(q.Wd && q.Rk && q.part && rank_ok(q.C, r, q.Wd, q.ldw, q.Rk, q.ldr, q.part, q.rows_per_block),
                       "cvft_lora_rank_partial_multi: problem %d: bf16 operands 16-byte aligned, widths %% 8 == 0, r in {16,32,48,64}", i);
        pp.p[i] = make_prob(q.C, r, q.Wd, q.ldw, q.Rk, q.ldr, q.part, q.transpose_out, q.rows_per_block);
    }
    for (int i = n; i < 4; ++i) pp.p[i] = pp.p[0];
    rank_launch(M, r, pp, n, (hipStream_t)stream);
    CVFT_LAUNCH_CHECK("cvft_lora_rank_partial_multi");
    return 0;
}

// n slab products of rank r, each with its own row count (host array of descriptors), in ceil(n / 64) launches.
extern "C" int cvft_lora_rank_partial_batch(int r, int n, const cvft_rank_prob_m* probs, void* stream) {
    CVFT_CHECK_ARG((r == 16 || r == 32 || r == 48 || r == 64) && n >= 1 && probs, "cvft_lora_rank_partial_batch: r in {16,32,48,64}, n >= 1");
    for (int i = 0; i < n; ++i) {
        const cvft_rank_prob_m& q = probs[i];
        CVFT_CHECK_ARG(q.M > 0 && q.Wd && q.Rk && q.part && rank_ok(q.C, r, q.Wd, q.ldw, q.Rk, q.ldr, q.part, q.rows_per_block),
                       "cvft_lora_rank_partial_batch: problem %d: bf16 operands 16-byte aligned, widths %% 8 == 0, rows_per_block %% 32 == 0", i);
    }
    const size_t sm = (size_t)4 * 2 * (32 * 128 + 32 * r * 2);
    hipStream_t st = (hipStream_t)stream;
    for (int i0 = 0; i0 < n; i0 += CVFT_RANK_BATCH) {
        const int nb = min(CVFT_RANK_BATCH, n - i0);
        RankBatch bt;
        int gx = 1, gy = 1;
        for (int i = 0; i < CVFT_RANK_BATCH; ++i) {
            const cvft_rank_prob_m& q = probs[i0 + (i < nb ? i : 0)];
            RankProbM& e = bt.p[i];
            e.Wd = (const bf16_t*)q.Wd; e.Rk = (const bf16_t*)q.Rk; e.out = q.part; e.M = q.M; e.Cn = q.C; e.ldw = q.ldw; e.ldr = q.ldr;
            e.rows_per_block = q.rows_per_block; e.transpose_out = q.transpose_out;
            if (i < nb) {
                gx = max(gx, q.rows_per_block % 128 == 0 ? (q.C + 63) / 64 : (q.C + 255) / 256);
                gy = max(gy, (q.M + q.rows_per_block - 1) / q.rows_per_block);
            }
        }
        CVFT_CHECK_ARG(gy <= 65535, "cvft_lora_rank_partial_batch: too many row blocks");
        dim3 grid(gx, gy, nb);
#define RMB_LAUNCH(RBv)                                                                                                  \
    do {                                                                                                                 \
        auto kern = lora_rank_mfma_batch_kernel<RBv>;                                                                    \
        static bool attr_set = false;                                                                                    \
        if (sm > 48 * 1024 && !attr_set) {                                                                               \
            attr_set = true;                                                                                             \
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); \
        }                                                                                                                \
        hipLaunchKernelGGL(kern, grid, dim3(256), sm, st, bt);                                                           \
    } while (0)
        if (r == 16) RMB_LAUNCH(1); else if (r == 32) RMB_LAUNCH(2); else if (r == 48) RMB_LAUNCH(3); else RMB_LAUNCH(4);
#undef RMB_LAUNCH
    }
    CVFT_LAUNCH_CHECK("cvft_lora_rank_partial_batch");
    return 0;
}
