// block_qkv.hip -- first half of the estimator's transformer block (matcha transformer.py:255-289 == modules.py:349-361) as
// row-tile chain kernels (structure and helpers: block_fused.hip / block_common.h):
//
//   cvft_block_qkv_fwd:  y = norm1(x);  Y = y Wqkv^T + b + (s / (1-p) sum_t drop_t(y) A_t^T) B_blk^T           (one launch)
//                        -- LayerNorm, the three LoRA adapters' rank-16 side products under lora_dropout (lora.py:70-73) and the
//                        stacked q|k|v projection; y never touches HBM (only the dropped copies the adapter gradients need).
//   cvft_block_qkv_bwd:  V = s dY B_blk;  dy = dY Wqkv + sum_t keep_t / (1-p) (V_t A_t);  dx = dres + norm1'(dy)  (one launch)
//
// The q|k|v weight arrives as a per-wave stream of pre-packed MFMA A-operand fragments (ring of 32); the adapters change every
// optimiser step, so their operands are read straight from the optimiser's bf16 shadows (a few KB, fragment-shaped loads).
// Masks: the counter-based masks of common.h (same (seed, site, element index) -> same mask as cvft_skinny_dropout / the masked
// rank extension of cvft_gemm), so the launch-per-stage form and this one agree mask for mask.
#include "block_qkv_body.h"

template <bool DROP, int NS>
__global__ __launch_bounds__(256, 1) void block_qkv_fwd_kernel(QkvFwd a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * BF_ROWS;
    const int row = min(m0 + m, a.M - 1);
    const bool rvalid = m0 + m < a.M;
    const int half = NS > 1 ? blockIdx.y : 0;

    // ---- loads in the order they are needed: x, small parameters (-> LDS), adapter operands, then the ring
    bf16x4 xb[2][4];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int g = 0; g < 4; ++g) xb[c2][g] = *reinterpret_cast<const bf16x4*>(a.x + (size_t)row * BF_D + 64 * wave + 32 * c2 + 8 * g + 4 * h);
    {
        BqParams pr;
        bq_params_load(a, wave, lane, pr);
        bq_params_store(a, smem, wave, lane, pr);
    }
    const int nt0 = (48 / NS) * half + (12 / NS) * wave;     // this wave's first n-tile
    BqOps<NS> ops;
    bq_load_ops<NS>(a, wave, m, h, nt0, ops);
    // (the packed stream is [48 n-tiles][16 k-steps] in tile order: tile nt0 starts at fragment 16 nt0)
    const bf16x8* nx = a.Wst + (size_t)nt0 * BF_KS * 64 + lane;
    bf16x8 ring[BF_RING];
#pragma unroll
    for (int i = 0; i < BF_RING; ++i) ring[i] = nx[i * 64];
    nx += BF_RING * 64;
    BfTouch touched;
    if (BF_TOUCH) touched = bf_touch_stream(a.Wst, 4 * a.wave_frags);
    bq_head_fwd<DROP, NS>(a, smem, xb, ring, nx, ops, lane, wave, row, rvalid, nt0, half == 0);
    if (BF_TOUCH && bf_touch_fold(touched) == 0x7fc07fc1u && a.M < 0) a.mean[0] = 0.f;      // (keeps the prefetch loads alive; never true)
}

template <bool DROP>
__global__ __launch_bounds__(256, 1) void block_qkv_bwd_kernel(QkvBwd a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16x8 ring[BF_RING];
    const bf16x8* nx;
    bf16x4 dxo[2][4];
    bq_bwd_body<DROP, false>(a, smem, ring, nx, dxo);
}

template <bool DROP, int NS>
static int launch_qkv_fwd(const QkvFwd& a, hipStream_t st) {
    static int ready = 0;
    if (!ready) { if (bf_prepare(block_qkv_fwd_kernel<DROP, NS>)) return -2; ready = 1; }
    hipLaunchKernelGGL((block_qkv_fwd_kernel<DROP, NS>), dim3((a.M + BF_ROWS - 1) / BF_ROWS, NS), dim3(256), BF_LDS_TOTAL, st, a);
    return 0;
}
template <bool DROP>
static int launch_qkv_bwd(const QkvBwd& a, hipStream_t st) {
    static int ready = 0;
    if (!ready) { if (bf_prepare(block_qkv_bwd_kernel<DROP>)) return -2; ready = 1; }
    hipLaunchKernelGGL((block_qkv_bwd_kernel<DROP>), dim3((a.M + BF_ROWS - 1) / BF_ROWS), dim3(256), BF_LDS_TOTAL, st, a);
    return 0;
}

int block_qkv_wide_fwd_launch(const cvft_block_qkv_args* p, void* stream);          // block_qkv_wide.hip
int block_qkv_wide_bwd_launch(const cvft_block_qkv_bwd_args* p, void* stream);

extern "C" int cvft_block_qkv_fwd(const cvft_block_qkv_args* p, void* stream) {
    CVFT_CHECK_ARG(p && p->M > 0 && p->N3 == 1536, "cvft_block_qkv_fwd: need M > 0 and 3N == 1536 (N3=%d)", p ? p->N3 : -1);
    CVFT_CHECK_ARG(p->x && p->gamma && p->beta && p->mean && p->rstd && p->W_fwd && p->A && p->Bb && p->U && p->Y, "cvft_block_qkv_fwd: null operand");
    CVFT_CHECK_ARG(cvft_drop_rate_ok(p->p) && (p->p == 0.f || p->seed), "cvft_block_qkv_fwd: bad dropout args (p == 0 or 2^-16 <= p <= 1 - 2^-16, seed)");
    CVFT_CHECK_ARG(al16(p->x) && al16(p->gamma) && al16(p->beta) && al16(p->W_fwd) && al16(p->A) && al16(p->Bb) && al16(p->Y) && (!p->bias || al16(p->bias)) &&
                   p->lda % 8 == 0 && p->ldb % 4 == 0 && p->ldu % 4 == 0 && p->ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(p->U) & 7) == 0 &&
                   (!p->y_out || al16(p->y_out)) && (!p->xd[0] || al16(p->xd[0])) && (!p->xd[1] || al16(p->xd[1])) && (!p->xd[2] || al16(p->xd[2])),
                   "cvft_block_qkv_fwd: operands must be 16-byte aligned (row pitches: lda %% 8, ldb / ldu / ldy %% 4)");
    if (p->wide) {
        CVFT_CHECK_ARG(p->ldy % 8 == 0, "cvft_block_qkv_fwd: the wide form stores Y in 16-byte pieces (ldy %% 8 == 0)");
        const int rc = block_qkv_wide_fwd_launch(p, stream);
        if (rc) return rc;
        CVFT_LAUNCH_CHECK("cvft_block_qkv_fwd (wide)");
        return 0;
    }
    QkvFwd a;
    a.M = p->M; a.x = (const bf16_t*)p->x; a.gamma = p->gamma; a.beta = p->beta; a.eps = p->eps; a.mean = p->mean; a.rstd = p->rstd;
    a.Wst = (const bf16x8*)p->W_fwd; a.wave_frags = p->N3 / 8; a.bias = p->bias; a.N3 = p->N3;
    a.A = (const bf16_t*)p->A; a.lda = p->lda; a.Bb = (const bf16_t*)p->Bb; a.ldb = p->ldb;
    a.alpha = p->alpha; a.p = p->p; a.seed = (const long long*)p->seed;
    for (int i = 0; i < 3; ++i) { a.sites[i] = p->sites[i]; a.xd[i] = (bf16_t*)p->xd[i]; }
    a.U = (bf16_t*)p->U; a.ldu = p->ldu; a.y_out = (bf16_t*)p->y_out; a.Y = (bf16_t*)p->Y; a.ldy = p->ldy;
    // two workgroups per row tile while the launch cannot fill the chip with one AND this chain is not competing with two others:
    // same-box A/B, flow_only (2 chains) 14.95 -> 14.71 ms, joint (3 chains) 23.37 -> 23.53 -- what shortens a chain on an idle chip
    // costs the step when the chip is shared.  CVFT_QKV_NSPLIT=1/2 forces.
    static const int ns_env = getenv("CVFT_QKV_NSPLIT") ? atoi(getenv("CVFT_QKV_NSPLIT")) : 0;
    const int ns = ns_env ? ns_env : (((p->M + BF_ROWS - 1) / BF_ROWS <= 128 && cvft_concurrent_chains() < 3) ? 2 : 1);
    hipStream_t st = (hipStream_t)stream;
    const int rc = p->p > 0.f ? (ns == 2 ? launch_qkv_fwd<true, 2>(a, st) : launch_qkv_fwd<true, 1>(a, st))
                              : (ns == 2 ? launch_qkv_fwd<false, 2>(a, st) : launch_qkv_fwd<false, 1>(a, st));
    if (rc) return rc;
    CVFT_LAUNCH_CHECK("cvft_block_qkv_fwd");
    return 0;
}

extern "C" int cvft_block_qkv_bwd(const cvft_block_qkv_bwd_args* p, void* stream) {
    CVFT_CHECK_ARG(p && p->M > 0 && p->N3 == 1536, "cvft_block_qkv_bwd: need M > 0 and 3N == 1536");
    CVFT_CHECK_ARG(p->dY && p->x && p->gamma && p->mean && p->rstd && p->W_bwd && p->At && p->Bbt && p->V && p->dx, "cvft_block_qkv_bwd: null operand");
    CVFT_CHECK_ARG(cvft_drop_rate_ok(p->p) && (p->p == 0.f || p->seed), "cvft_block_qkv_bwd: bad dropout args (p == 0 or 2^-16 <= p <= 1 - 2^-16, seed)");
    CVFT_CHECK_ARG(al16(p->dY) && al16(p->x) && al16(p->gamma) && al16(p->W_bwd) && al16(p->Bbt) && al16(p->dx) && (!p->dres || al16(p->dres)) &&
                   p->lddy % 8 == 0 && p->ldbt % 8 == 0 && p->ldat % 4 == 0 && p->ldv % 4 == 0 && (reinterpret_cast<uintptr_t>(p->At) & 7) == 0 &&
                   (reinterpret_cast<uintptr_t>(p->V) & 7) == 0, "cvft_block_qkv_bwd: operands must be 16-byte aligned (row pitches: lddy / ldbt %% 8, ldat / ldv %% 4)");
    if (p->wide) {
        const int rc = block_qkv_wide_bwd_launch(p, stream);
        if (rc) return rc;
        CVFT_LAUNCH_CHECK("cvft_block_qkv_bwd (wide)");
        return 0;
    }
    QkvBwd a;
    a.M = p->M; a.dY = (const bf16_t*)p->dY; a.lddy = p->lddy; a.dres = (const bf16_t*)p->dres; a.x = (const bf16_t*)p->x;
    a.gamma = p->gamma; a.mean = p->mean; a.rstd = p->rstd; a.Wst = (const bf16x8*)p->W_bwd; a.wave_frags = p->N3 / 8; a.N3 = p->N3;
    a.At = (const bf16_t*)p->At; a.ldat = p->ldat; a.Bbt = (const bf16_t*)p->Bbt; a.ldbt = p->ldbt;
    a.alpha = p->alpha; a.p = p->p; a.seed = (const long long*)p->seed;
    for (int i = 0; i < 3; ++i) a.sites[i] = p->sites[i];
    a.V = (bf16_t*)p->V; a.ldv = p->ldv; a.dx = (bf16_t*)p->dx;
    const int rc = p->p > 0.f ? launch_qkv_bwd<true>(a, (hipStream_t)stream) : launch_qkv_bwd<false>(a, (hipStream_t)stream);
    if (rc) return rc;
    CVFT_LAUNCH_CHECK("cvft_block_qkv_bwd");
    return 0;
}
