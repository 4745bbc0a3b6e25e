// block_fused.hip -- the estimator's transformer block (matcha transformer.py:243-316 == modules.py:296-375) as row-tile
// kernels: a workgroup owns 32 rows of the [batch*time, 256] residual stream and carries them through a whole CHAIN of
// linears, so the [M, 1024] hidden activations and the LayerNorm outputs never touch HBM.
//
//   cvft_block_tail_fwd:  x1 = x0 + o Wo^T + bo ;  out = x1 + W2 act(W1 LN(x1) + b1) + b2          (one launch)
//   cvft_block_tail_bwd:  dx1 = dy + LN'(W1^T (act'(z) . (W2^T dy))) ;  do = dx1 Wo                 (one launch)
//
// Structure (MI355X): d = 256 is small, so a row tile's activations fit in ONE wave's registers as MFMA B-operand
// fragments (32 rows x 256 = 64 VGPRs), and every product is computed TRANSPOSED -- D^T[feature, row] = W[feature, k] .
// X^T[k, row], weights as the A operand -- so that
//   * the frozen weights are pre-packed on the host in MFMA A-fragment order and IN THE ORDER THE WAVE CONSUMES THEM
//     (hipops/blockpack.py): a wave's weights are one linear stream of 1-KB fragments loaded straight into VGPRs (no LDS
//     image, no barrier, no bank conflicts: the weights of a row tile are read once per wave and never shared --
//     cdna_hip_programming.md section 5, "GEMV / M <= 16 ... neither").  The stream runs through a ring of 32 fragment
//     registers; every register is re-requested for stream position p + 32 right behind the MFMA that consumed position p,
//     so 32 KB per wave stay in flight through every phase of the kernel, barriers and LayerNorm included;
//   * the hidden tile a wave has just produced (accumulator layout: row on the lane, feature in the register index) is,
//     after activation and bf16 packing, directly the B operand of the next product (cdna_hip_programming.md section 3,
//     "An accumulator tile as the next MFMA's operand"); the k permutation that costs is folded into the packed weights.
// The 4 waves of a workgroup split the REDUCTION dimension of each chain link (hidden units of the feed-forward, input
// features of the output projection) and meet once per link in LDS (fp32 partial tiles, 128 KB).
// Bound: the per-CU L2 -> register rate (~120 GB/s per CU with one workgroup per CU, tools/ub/mfma_rate.hip): 1.25 MB of
// packed weights per row tile each way; 125 workgroups at M = 4000.
#include "block_qkv_body.h"

struct TailFwd {
    int M;
    const bf16_t* o; int ldo;
    const bf16_t* x0;
    const bf16x8* Wst; int wave_frags;
    const float* bo;
    bf16_t* x1;
    const float* gamma; const float* beta; float eps;
    const float* b1; int F;
    const float* b2;
    bf16_t* z;
    float* mean; float* rstd;
    bf16_t* out;
};

// W2 product of one hidden tile: z (accumulator layout, bias included) -> save, GELU, pack -> acc2 += W2 fragments (ring slots
// S0 .. S0+15, order [s][ct]) . h;  RELOAD: re-request the 16 slots for stream positions +32 (nx points at this product's
// first fragment + 32)
template <int ACT, int S0, bool RELOAD>
__device__ __forceinline__ void bf_ffn_second(bf16x8 (&ring)[BF_RING], const bf16x8* nx, const f32x16& z, f32x16 (&acc2)[BF_CT], bf16x8* zdst) {
    bf16x8 hb[2];
    if (zdst != nullptr) {                             // training: the workspace takes gelu'(z) (block_common.h, bf_gelu2)
        bf16x8 zs[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float g, dg;
            bf_gelu2<ACT>(z[i], g, dg);
            zs[i >> 3][i & 7] = (bf16_t)dg;
            hb[i >> 3][i & 7] = (bf16_t)g;
        }
        zdst[0] = zs[0];
        zdst[1] = zs[1];
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float g, dg;
            bf_gelu2<ACT>(z[i], g, dg);
            hb[i >> 3][i & 7] = (bf16_t)g;
        }
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int ct = 0; ct < BF_CT; ++ct) {
            acc2[ct] = mfma32(ring[S0 + s2 * 8 + ct], hb[s2], acc2[ct]);
            if (RELOAD) ring[S0 + s2 * 8 + ct] = nx[(s2 * 8 + ct) * 64];
        }
}

// AR = rounds of 32 fragments in the output projection's stream (DI / 256; 0 = no projection, x1 is the input)
// LINK: the NEXT block's head (norm1 + LoRA q|k|v projection, block_qkv_body.h: q) follows on the same 32 rows in the same launch:
// the block output stays in the wave's registers, and a.Wst is the linked stream -- per wave its tail fragments, then its 192
// fragments of the head (hipops/blockpack.py, BlockLinkPack) -- so the ring's re-requests run from one product into the other
// without a refill: the last feed-forward round already requests the head's first 32 fragments.
template <int ACT, int AR, bool LINK, bool DROP>
__device__ __forceinline__ void bt_fwd_body(const TailFwd& a, const QkvFwd& q, char* smem) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * BF_ROWS;
    const int row = min(m0 + m, a.M - 1);              // clamped for loads; stores are predicated on rvalid
    const bool rvalid = m0 + m < a.M;
    BF_STAMP(0);

    // ---- loads in the order they are needed (vmcnt retires in order): activations, small parameters (-> LDS), then the ring
    bf16x8 of[AR > 0 ? 4 * AR : 1];
    bf16x4 xb[2][4];                                   // this wave's 64 features of x0, later of x1 (as stored: bf16)
    if (AR > 0) {
        const bf16_t* op = a.o + (size_t)row * a.ldo + wave * (64 * AR) + 8 * h;
#pragma unroll
        for (int k = 0; k < 4 * AR; ++k) of[k] = *reinterpret_cast<const bf16x8*>(op + 16 * k);
    }
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            xb[c2][g] = *reinterpret_cast<const bf16x4*>((AR > 0 ? a.x0 : a.x1) + (size_t)row * BF_D + 64 * wave + 32 * c2 + 8 * g + 4 * h);
    {
        f32x4* par = reinterpret_cast<f32x4*>(smem + BF_LDS_PAR);
        const int i = threadIdx.x & 63;                // wave w stages parameter w: 256 floats = 64 x 16 B
        const float* srcp = wave == 0 ? a.bo : wave == 1 ? a.gamma : wave == 2 ? a.beta : a.b2;
        if (srcp != nullptr) par[wave * 64 + i] = reinterpret_cast<const f32x4*>(srcp)[i];
        for (int k = threadIdx.x; k < a.F / 4; k += 256)
            reinterpret_cast<f32x4*>(smem + BF_LDS_BIAS)[k] = reinterpret_cast<const f32x4*>(a.b1)[k];
    }
    // the wave's weight stream: fill the ring, `nx` = address of stream position (next to consume) + 32
    const bf16x8* nx = a.Wst + (size_t)wave * a.wave_frags * 64 + lane;
    bf16x8 ring[BF_RING];
#pragma unroll
    for (int i = 0; i < BF_RING; ++i) ring[i] = nx[i * 64];
    nx += BF_RING * 64;
    BfTouch touched;
    if (BF_TOUCH && !LINK) touched = bf_touch_stream(a.Wst, 4 * a.wave_frags);
    if (BF_TOUCH && LINK) bf_touch_stream_lds(a.Wst, 4 * a.wave_frags, smem + BF_LDS_DUMMY);
    const float* pbo = reinterpret_cast<const float*>(smem + BF_LDS_PAR), *pgam = pbo + BF_D, *pbet = pbo + 2 * BF_D, *pb2 = pbo + 3 * BF_D;

    if (AR > 0) {
        // ---- x1 = x0 + o Wo^T + bo: wave w reduces over input features [w*DI/4, (w+1)*DI/4); stream order [ks][ct]
        f32x16 acc[BF_CT];
#pragma unroll
        for (int ct = 0; ct < BF_CT; ++ct) acc[ct] = zero16();
#pragma unroll
        for (int r = 0; r < AR; ++r) {
#pragma unroll
            for (int j = 0; j < BF_RING; ++j) {
                acc[j & 7] = mfma32(ring[j], of[4 * r + (j >> 3)], acc[j & 7]);
                ring[j] = nx[j * 64];
            }
            nx += BF_RING * 64;
        }
        BF_STAMP(1);
        float v[2][16];
        bf_reduce(smem, wave, lane, acc, v);
        BF_STAMP(2);
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 64 * wave + 32 * c2 + 8 * g + 4 * h;
                const f32x4 bb = *reinterpret_cast<const f32x4*>(pbo + c);
#pragma unroll
                for (int i = 0; i < 4; ++i) xb[c2][g][i] = (bf16_t)(v[c2][4 * g + i] + bb[i] + (float)xb[c2][g][i]);
                if (rvalid) *reinterpret_cast<bf16x4*>(a.x1 + (size_t)row * BF_D + c) = xb[c2][g];
            }
    } else {
        __syncthreads();                               // (the staged parameters)
    }
    BF_STAMP(3);
    // ---- LayerNorm (two-pass fp32 statistics, like ln_fwd_kernel) -> y tile in LDS -> every wave's B fragments
    float s = 0.f;
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) s += (float)xb[c2][g][i];
    const float mean = bf_rowsum(smem, 0, wave, lane, s) * (1.f / BF_D);
    s = 0.f;
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float d = (float)xb[c2][g][i] - mean; s += d * d; }
    const float rstd = rsqrtf(bf_rowsum(smem, 1, wave, lane, s) * (1.f / BF_D) + a.eps);
    if (wave == 0 && lane < 32 && rvalid) { a.mean[row] = mean; a.rstd[row] = rstd; }
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 64 * wave + 32 * c2 + 8 * g + 4 * h;
            const f32x4 gg = *reinterpret_cast<const f32x4*>(pgam + c);
            const f32x4 be = *reinterpret_cast<const f32x4*>(pbet + c);
            bf16x4 y;
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = (bf16_t)(((float)xb[c2][g][i] - mean) * rstd * gg[i] + be[i]);
            *reinterpret_cast<bf16x4*>(smem + BF_LDS_TILE + bf_tile_off(m, c)) = y;
        }
    __syncthreads();
    bf16x8 yf[BF_KS];
    bf_tile_read(smem, lane, yf);
    BF_STAMP(4);

    // ---- feed-forward: wave w owns hidden tiles [w*F/128, (w+1)*F/128); acc2 = its share of W2 act(.)
    // stream: W1(0), then per tile t: W1(t+1) (slots 16..31), W2(t) (slots 0..15); the last tile's W2 sits in slots 16..31.
    // The W1 product of tile t+1 (matrix core) is issued in front of the GELU of tile t (VALU): one wave per SIMD has nobody
    // else to overlap them with.
    const int ntw = a.F / 128;
    const int ht0 = wave * ntw;
    const float* b1s = reinterpret_cast<const float*>(smem + BF_LDS_BIAS);
    f32x16 acc2[BF_CT];
#pragma unroll
    for (int ct = 0; ct < BF_CT; ++ct) acc2[ct] = zero16();
    f32x16 acc1 = bf_bias_init(b1s, ht0, h);
#pragma unroll
    for (int ks = 0; ks < BF_KS; ++ks) {
        acc1 = mfma32(ring[ks], yf[ks], acc1);
        ring[ks] = nx[ks * 64];
    }
    nx += BF_KS * 64;
    bf16x8* zp = a.z == nullptr ? nullptr : reinterpret_cast<bf16x8*>(a.z) + ((size_t)(blockIdx.x * (a.F / 32) + ht0) * 64 + lane) * 2;
    BF_STAMP(5);
    for (int t = 0; t + 1 < ntw; ++t) {
        BF_STAMP(8 + t);
        const f32x16 z = acc1;
        acc1 = bf_bias_init(b1s, ht0 + t + 1, h);
#pragma unroll
        for (int ks = 0; ks < BF_KS; ++ks) {
            acc1 = mfma32(ring[16 + ks], yf[ks], acc1);
            ring[16 + ks] = nx[ks * 64];
        }
        bf_ffn_second<ACT, 0, true>(ring, nx + 16 * 64, z, acc2, zp == nullptr ? nullptr : zp + (size_t)t * 128);
        nx += BF_RING * 64;
    }
    BF_STAMP(8 + ntw - 1);
    // LINK: the head's LayerNorm parameters / bias are requested here, in front of the last round's re-requests (vmcnt retires in
    // order: they arrive first), and the last round re-requests its 16 slots too -- positions 16..31 of the head's stream (slots
    // 0..15 took positions 0..15 in the round before)
    BqParams hpar;
    if (LINK) bq_params_load(q, wave, lane, hpar);
    bf_ffn_second<ACT, 16, LINK>(ring, nx, acc1, acc2, zp == nullptr ? nullptr : zp + (size_t)(ntw - 1) * 128);
    if (LINK) nx += 16 * 64;
    float v[2][16];
    BF_STAMP(6);
    __syncthreads();                                   // (the y tile / statistics reads are done before `part` is rewritten)
    // LINK: lane and row re-enter through an empty asm -- everything the rest of the kernel derives from them (the reduction's LDS
    // addresses, the output rows, some thirty address registers of the head) is then computed HERE and not at the top of the kernel,
    // where it would sit through the feed-forward loop and spill; a spill reload is a memory load, i.e. a wait for the whole
    // fragment ring in flight
    int lane2 = lane, row2 = row;
    if (LINK) asm volatile("" : "+v"(lane2), "+v"(row2));
    const int h2 = lane2 >> 5;
    if (LINK) bq_params_store(q, smem, wave, lane2, hpar);  // (b1 / gamma / beta were last read in front of that barrier; b2's slot stays)
    bf_reduce(smem, wave, lane2, acc2, v);
    BF_STAMP(7);
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 64 * wave + 32 * c2 + 8 * g + 4 * h2;
            const f32x4 bb = *reinterpret_cast<const f32x4*>(pb2 + c);
            bf16x4 yo;
#pragma unroll
            for (int i = 0; i < 4; ++i) yo[i] = (bf16_t)(v[c2][4 * g + i] + bb[i] + (float)xb[c2][g][i]);
            if (rvalid) *reinterpret_cast<bf16x4*>(a.out + (size_t)row2 * BF_D + c) = yo;
            if (LINK) xb[c2][g] = yo;                  // the next block's input, as stored
        }
    if (LINK) {
        BqOps<1> hops;                                 // (80 registers: requested only now that the tail's accumulators are dead; first
        bq_load_ops<1>(q, wave, lane2 & 31, lane2 >> 5, 12 * wave, hops);   //  used behind the LayerNorm's three barriers)
        bq_head_fwd<DROP, 1>(q, smem, xb, ring, nx, hops, lane2, wave, row2, rvalid, 12 * wave, true);
    }
    if (BF_TOUCH && !LINK && bf_touch_fold(touched) == 0x7fc07fc1u && a.M < 0) a.mean[0] = 0.f;      // (keeps the prefetch loads alive; never true)
}

template <int ACT, int AR>
__global__ __launch_bounds__(256, 1) void block_tail_fwd_kernel(TailFwd a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    QkvFwd none;
    bt_fwd_body<ACT, AR, false, false>(a, none, smem);
}
template <int ACT, int AR, bool DROP>
__global__ __launch_bounds__(256, 1) void block_link_fwd_kernel(TailFwd a, QkvFwd q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bt_fwd_body<ACT, AR, true, DROP>(a, q, smem);
}

struct TailBwd {
    int M;
    const bf16_t* x1; const bf16_t* dy;
    const float* gamma; const float* mean; const float* rstd;
    const bf16_t* z;
    const bf16x8* Wst; int wave_frags; int F;
    bf16_t* dx1;
    bf16_t* dout; int lddo;
    const bf16_t* ao; const bf16_t* ao_lo; int ldao; float* delta; int T;      // (optional: delta for the attention backward, cvft.h)
};

// W1^T product of one hidden tile: g = W2^T dy (accumulator layout), zs = the tile's saved pre-activations -> dz = g act'(z),
// packed -> accd += W1^T fragments (ring slots S0 .. S0+15, order [s][dt]) . dz
template <int ACT, int S0, bool RELOAD>
__device__ __forceinline__ void bf_ffn_second_bwd(bf16x8 (&ring)[BF_RING], const bf16x8* nx, const f32x16& g, const bf16x8 (&zs)[2], f32x16 (&accd)[BF_CT]) {
    bf16x8 hb[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) hb[i >> 3][i & 7] = (bf16_t)(g[i] * (float)zs[i >> 3][i & 7]);      // (the workspace holds gelu'(z))
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int ct = 0; ct < BF_CT; ++ct) {
            accd[ct] = mfma32(ring[S0 + s2 * 8 + ct], hb[s2], accd[ct]);
            if (RELOAD) ring[S0 + s2 * 8 + ct] = nx[(s2 * 8 + ct) * 64];
        }
}

// CR = rounds of 32 fragments in the output projection's dgrad stream (DI / 256; 0 = none)
// LINK (block_link_bwd_kernel): the NEXT block's head backward ran in front on the same rows -- dy arrives in registers (dyw: this
// wave's 64 features of the 32 rows, as stored), the ring already holds this wave's first 32 fragments of the tail's stream and nx
// points behind them.
template <int ACT, int CR, bool LINK>
__device__ __forceinline__ void bt_bwd_body(const TailBwd& a, char* smem, bf16x8 (&ring)[BF_RING], const bf16x8* nx, const bf16x4 (&dyw)[2][4]) {
    int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // (LINK: the lane re-enters through an empty asm, so that this half's addresses are computed here and not in front of the head's
    //  loop, where they would spill -- block_link_fwd_kernel has the story)
    if (LINK) asm volatile("" : "+v"(lane));
    const int m = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * BF_ROWS;
    const int row = min(m0 + m, a.M - 1);
    const bool rvalid = m0 + m < a.M;

    // ---- loads in the order they are needed (vmcnt retires in order): dy fragments, the first tile's pre-activations, the
    // LayerNorm backward's operands (used at the end: requested now so that they never queue behind the ring), then the ring
    // dy as B fragments (natural k order): lane (m, h) holds dy[row][16 ks + 8 h .. + 7]
    bf16x8 dyf[BF_KS];
    if (!LINK) {
#pragma unroll
        for (int ks = 0; ks < BF_KS; ++ks) dyf[ks] = *reinterpret_cast<const bf16x8*>(a.dy + (size_t)row * BF_D + 16 * ks + 8 * h);
    } else {
        // through the LDS tile: every wave needs all 256 features of its rows as B fragments
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<bf16x4*>(smem + BF_LDS_TILE + bf_tile_off(m, 64 * wave + 32 * c2 + 8 * g + 4 * h)) = dyw[c2][g];
    }
    const int ntw = a.F / 128;
    const bf16x8* zp = reinterpret_cast<const bf16x8*>(a.z) + ((size_t)(blockIdx.x * (a.F / 32) + wave * ntw) * 64 + lane) * 2;
    bf16x8 zs[2] = {zp[0], zp[1]};
    bf16x4 xr[2][4], dr[2][4];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 64 * wave + 32 * c2 + 8 * g + 4 * h;
            xr[c2][g] = *reinterpret_cast<const bf16x4*>(a.x1 + (size_t)row * BF_D + c);
            dr[c2][g] = LINK ? dyw[c2][g] : *reinterpret_cast<const bf16x4*>(a.dy + (size_t)row * BF_D + c);
        }
    const float mean = a.mean[row], rstd = a.rstd[row];
    // (LINK: slot 0 held the head's gamma, last read in front of the head's two row-sum barriers)
    if (wave == 0) reinterpret_cast<f32x4*>(smem + BF_LDS_PAR)[lane] = reinterpret_cast<const f32x4*>(a.gamma)[lane];
    const float* pgam = reinterpret_cast<const float*>(smem + BF_LDS_PAR);
    BfTouch touched;
    if (!LINK) {
        nx = a.Wst + (size_t)wave * a.wave_frags * 64 + lane;
#pragma unroll
        for (int i = 0; i < BF_RING; ++i) ring[i] = nx[i * 64];
        nx += BF_RING * 64;
        if (BF_TOUCH) touched = bf_touch_stream(a.Wst, 4 * a.wave_frags);
    } else {
        __syncthreads();
        bf_tile_read(smem, lane, dyf);
    }

    // stream: W2^T(0), then per tile t: W2^T(t+1) (slots 16..31), W1^T(t) (slots 0..15); the last tile's W1^T in slots 16..31;
    // then the output projection's dgrad (CR rounds of 32)
    f32x16 accd[BF_CT];
#pragma unroll
    for (int ct = 0; ct < BF_CT; ++ct) accd[ct] = zero16();
    f32x16 accg = zero16();
#pragma unroll
    for (int ks = 0; ks < BF_KS; ++ks) {
        accg = mfma32(ring[ks], dyf[ks], accg);
        ring[ks] = nx[ks * 64];
    }
    nx += BF_KS * 64;
    for (int t = 0; t + 1 < ntw; ++t) {
        const f32x16 g = accg;
        const bf16x8 zc[2] = {zs[0], zs[1]};
        zs[0] = zp[(size_t)(t + 1) * 128];
        zs[1] = zp[(size_t)(t + 1) * 128 + 1];
        accg = zero16();
#pragma unroll
        for (int ks = 0; ks < BF_KS; ++ks) {
            accg = mfma32(ring[16 + ks], dyf[ks], accg);
            ring[16 + ks] = nx[ks * 64];
        }
        bf_ffn_second_bwd<ACT, 0, true>(ring, nx + 16 * 64, g, zc, accd);
        nx += BF_RING * 64;
    }
    bf_ffn_second_bwd<ACT, 16, (CR > 0)>(ring, nx, accg, zs, accd);
    nx += 16 * 64;
    float v[2][16];
    bf_reduce(smem, wave, lane, accd, v);              // v = dL/dy_norm for this wave's 64 features
    // ---- LayerNorm backward + residual branch: dx1 = dy + rstd (g.v - mean_c(g.v) - xhat mean_c(g.v.xhat))
    float xh[2][16], gv[2][16];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 gg = *reinterpret_cast<const f32x4*>(pgam + 64 * wave + 32 * c2 + 8 * g + 4 * h);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 4 * g + i;
                xh[c2][e] = ((float)xr[c2][g][i] - mean) * rstd;
                gv[c2][e] = gg[i] * v[c2][e];
                s1 += gv[c2][e];
                s2 += gv[c2][e] * xh[c2][e];
            }
        }
    const float m1 = bf_rowsum(smem, 0, wave, lane, s1) * (1.f / BF_D);
    const float m2 = bf_rowsum(smem, 1, wave, lane, s2) * (1.f / BF_D);
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 64 * wave + 32 * c2 + 8 * g + 4 * h;
            bf16x4 dx;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 4 * g + i;
                dx[i] = (bf16_t)((float)dr[c2][g][i] + rstd * (gv[c2][e] - m1 - xh[c2][e] * m2));
            }
            if (rvalid) *reinterpret_cast<bf16x4*>(a.dx1 + (size_t)row * BF_D + c) = dx;
            if (CR > 0) *reinterpret_cast<bf16x4*>(smem + BF_LDS_TILE + bf_tile_off(m, c)) = dx;
        }
    if (BF_TOUCH && !LINK && bf_touch_fold(touched) == 0x7fc07fc1u && a.M < 0) a.dx1[0] = (bf16_t)0.f;  // (keeps the prefetch loads alive; never true)
    if (CR == 0) return;
    // delta for the attention backward that consumes dout (optional, cvft.h)
    const bool wdel = a.delta != nullptr;
    const bf16_t* aol = a.ao_lo ? a.ao_lo : a.ao;         // (no residual: the same values, weighted 0 -- unconditional loads)
    const float lo_w = a.ao_lo ? 1.f : 0.f;
    __syncthreads();
    // ---- do = dx1 Wo: wave w owns output features [w*DI/4, (w+1)*DI/4): 2 feature tiles per round, stream order [ks][f2]
    bf16x8 dxf[BF_KS];
    bf_tile_read(smem, lane, dxf);
#pragma unroll
    for (int r = 0; r < CR; ++r) {
        // (the attention output's values of this lane's row and this round's 64 columns: requested in front of the round's products)
        bf16x4 ob[2][4], lb[2][4];
        if (wdel) {
#pragma unroll
            for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const size_t off = (size_t)row * a.ldao + 32 * (wave * 2 * CR + 2 * r + f2) + 8 * g + 4 * h;
                    ob[f2][g] = *reinterpret_cast<const bf16x4*>(a.ao + off);
                    lb[f2][g] = *reinterpret_cast<const bf16x4*>(aol + off);
                }
        }
        f32x16 acc[2] = {zero16(), zero16()};
#pragma unroll
        for (int j = 0; j < BF_RING; ++j) {
            acc[j & 1] = mfma32(ring[j], dxf[j >> 1], acc[j & 1]);
            if (r + 1 < CR) ring[j] = nx[j * 64];
        }
        nx += BF_RING * 64;
        float part = 0.f;
#pragma unroll
        for (int f2 = 0; f2 < 2; ++f2)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 d;
#pragma unroll
                for (int i = 0; i < 4; ++i) d[i] = (bf16_t)acc[f2][4 * g + i];
                if (rvalid) *reinterpret_cast<bf16x4*>(a.dout + (size_t)row * a.lddo + 32 * (wave * 2 * CR + 2 * r + f2) + 8 * g + 4 * h) = d;
                if (wdel) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) part += (float)d[i] * ((float)ob[f2][g][i] + lo_w * (float)lb[f2][g][i]);
                }
            }
        if (wdel) {      // round r of wave w is head w CR + r (64 columns = its two feature tiles); the two lane halves hold 16 + 16 of each tile
            part += __shfl_xor(part, 32);
            if (h == 0 && rvalid) a.delta[((size_t)(row / a.T) * (2 * CR) * 2 + (wave * CR + r)) * a.T + row % a.T] = part;
        }
    }
}

template <int ACT, int CR>
__global__ __launch_bounds__(256, 1) void block_tail_bwd_kernel(TailBwd a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16x8 ring[BF_RING];
    bf16x4 none[2][4];
    bt_bwd_body<ACT, CR, false>(a, smem, ring, nullptr, none);
}
// The head backward of block i + 1 (block_qkv_body.h, bq_bwd_body) and the tail backward of block i on the same 32 rows: the
// gradient at the block boundary stays in registers, q.Wst is the linked stream (per wave: 192 head fragments, then the tail's).
template <int ACT, int CR, bool DROP>
__global__ __launch_bounds__(256, 1) void block_link_bwd_kernel(QkvBwd q, TailBwd a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16x8 ring[BF_RING];
    const bf16x8* nx;
    bf16x4 dxo[2][4];
    bq_bwd_body<DROP, true>(q, smem, ring, nx, dxo);
    bt_bwd_body<ACT, CR, true>(a, smem, ring, nx, dxo);
}

template <int ACT, int AR>
static int launch_fwd(const TailFwd& a, hipStream_t st) {
    static int ready = 0;
    if (!ready) { if (bf_prepare(block_tail_fwd_kernel<ACT, AR>)) return -2; ready = 1; }
    hipLaunchKernelGGL((block_tail_fwd_kernel<ACT, AR>), dim3((a.M + BF_ROWS - 1) / BF_ROWS), dim3(256), BF_LDS_TOTAL, st, a);
    return 0;
}
template <int ACT, int CR>
static int launch_bwd(const TailBwd& a, hipStream_t st) {
    static int ready = 0;
    if (!ready) { if (bf_prepare(block_tail_bwd_kernel<ACT, CR>)) return -2; ready = 1; }
    hipLaunchKernelGGL((block_tail_bwd_kernel<ACT, CR>), dim3((a.M + BF_ROWS - 1) / BF_ROWS), dim3(256), BF_LDS_TOTAL, st, a);
    return 0;
}

int block_tail_lean_fwd_launch(const cvft_block_tail_args* p, int DI, void* stream);       // block_lean.hip
int block_tail_lean_bwd_launch(const cvft_block_tail_bwd_args* p, int DI, void* stream);
int block_tail_wide_fwd_launch(const cvft_block_tail_args* p, int DI, void* stream);       // block_wide.hip
int block_tail_wide_bwd_launch(const cvft_block_tail_bwd_args* p, int DI, void* stream);
int block_tail_wide8_fwd_launch(const cvft_block_tail_args* p, int DI, void* stream);      // block_wide8.hip

extern "C" int cvft_block_tail_fwd(const cvft_block_tail_args* p, void* stream) {
    CVFT_CHECK_ARG(p && p->M > 0 && p->F > 0 && p->F % 128 == 0 && p->F <= BF_MAX_F, "cvft_block_tail_fwd: need M > 0, F %% 128 == 0, F <= 2048 (F=%d)", p ? p->F : -1);
    CVFT_CHECK_ARG(p->x1 && p->out && p->W_fwd && p->b1 && p->b2 && p->gamma && p->beta && p->mean && p->rstd, "cvft_block_tail_fwd: null operand");
    CVFT_CHECK_ARG(p->act == CVFT_ACT_GELU_ERF || p->act == CVFT_ACT_GELU_TANH, "cvft_block_tail_fwd: act must be a GELU form");
    const int DI = p->o ? p->DI : 0;
    if (p->o) CVFT_CHECK_ARG(p->x0 && p->bo && (DI == 256 || DI == 512) && p->ldo % 8 == 0 && al16(p->o),
                             "cvft_block_tail_fwd: output projection needs x0, bo, DI in {256, 512}, 16-byte aligned o rows");
    CVFT_CHECK_ARG(al16(p->x1) && al16(p->out) && al16(p->W_fwd) && al16(p->b1) && al16(p->b2) && al16(p->gamma) && al16(p->beta) &&
                   (!p->z || al16(p->z)) && (!p->x0 || al16(p->x0)) && (!p->bo || al16(p->bo)), "cvft_block_tail_fwd: operands must be 16-byte aligned");
    if (p->lean) {
        CVFT_CHECK_ARG(p->lean == 1 || ((p->lean == 2 || p->lean == 4) && p->z && p->F >= 256),
                       "cvft_block_tail_fwd: lean must be 0, 1, 2 or 4; the 64-row forms (2, 4) need F >= 256 and always store z (whole 64-row groups)");
        CVFT_CHECK_ARG(p->lean != 4 || (p->F % 256 == 0 && p->F >= 512 && p->F <= 1024), "cvft_block_tail_fwd: the eight-wave form (4) needs F %% 256 == 0, 512 <= F <= 1024");
        const int rc = p->lean == 4 ? block_tail_wide8_fwd_launch(p, DI, stream)
                     : p->lean == 2 ? block_tail_wide_fwd_launch(p, DI, stream) : block_tail_lean_fwd_launch(p, DI, stream);
        if (rc) return rc;
        CVFT_LAUNCH_CHECK("cvft_block_tail_fwd (lean)");
        return 0;
    }
    TailFwd a;
    a.M = p->M; a.o = (const bf16_t*)p->o; a.ldo = p->ldo; a.x0 = (const bf16_t*)p->x0;
    a.Wst = (const bf16x8*)p->W_fwd; a.wave_frags = DI / 8 + p->F / 4; a.bo = p->bo; a.x1 = (bf16_t*)p->x1;
    a.gamma = p->gamma; a.beta = p->beta; a.eps = p->eps; a.b1 = p->b1; a.F = p->F; a.b2 = p->b2;
    a.z = (bf16_t*)p->z; a.mean = p->mean; a.rstd = p->rstd; a.out = (bf16_t*)p->out;
    const bool erf = p->act == CVFT_ACT_GELU_ERF;
    int rc;
    hipStream_t st = (hipStream_t)stream;
    if (DI == 0) rc = erf ? launch_fwd<CVFT_ACT_GELU_ERF, 0>(a, st) : launch_fwd<CVFT_ACT_GELU_TANH, 0>(a, st);
    else if (DI == 256) rc = erf ? launch_fwd<CVFT_ACT_GELU_ERF, 1>(a, st) : launch_fwd<CVFT_ACT_GELU_TANH, 1>(a, st);
    else rc = erf ? launch_fwd<CVFT_ACT_GELU_ERF, 2>(a, st) : launch_fwd<CVFT_ACT_GELU_TANH, 2>(a, st);
    if (rc) return rc;
    CVFT_LAUNCH_CHECK("cvft_block_tail_fwd");
    return 0;
}

template <int ACT, int AR, bool DROP>
static int launch_link_fwd(const TailFwd& a, const QkvFwd& q, hipStream_t st) {
    static int ready = 0;
    if (!ready) { if (bf_prepare(block_link_fwd_kernel<ACT, AR, DROP>, BF_LDS_TOTAL_DMA)) return -2; ready = 1; }
    hipLaunchKernelGGL((block_link_fwd_kernel<ACT, AR, DROP>), dim3((a.M + BF_ROWS - 1) / BF_ROWS), dim3(256), BF_LDS_TOTAL_DMA, st, a, q);
    return 0;
}

// The tail of one block and the head of the NEXT one on the same rows in one launch (include/cvft.h).  Both argument blocks mean
// what they mean for cvft_block_tail_fwd / cvft_block_qkv_fwd, with t->lean == 0, q->wide == 0 and q->x == t->out; W_link replaces
// both weight streams.
extern "C" int cvft_block_link_fwd(const cvft_block_tail_args* p, const cvft_block_qkv_args* q, const void* W_link, void* stream) {
    CVFT_CHECK_ARG(p && q && W_link && al16(W_link), "cvft_block_link_fwd: null / unaligned operand");
    CVFT_CHECK_ARG(p->M > 0 && q->M == p->M && p->F % 128 == 0 && p->F >= 256 && p->F <= 1024 && q->N3 == 1536,
                   "cvft_block_link_fwd: need equal M > 0, 256 <= F <= 1024 (F %% 128 == 0), 3N == 1536 (M=%d/%d F=%d)", p->M, q->M, p->F);
    CVFT_CHECK_ARG(p->lean == 0 && q->wide == 0 && p->o && p->DI == 512 && p->x0 && p->bo && p->ldo % 8 == 0 && al16(p->o),
                   "cvft_block_link_fwd: the 32-row forms only (lean == 0, wide == 0), with the output projection (DI == 512)");
    CVFT_CHECK_ARG(p->x1 && p->out && p->b1 && p->b2 && p->gamma && p->beta && p->mean && p->rstd && q->x == p->out &&
                   q->gamma && q->beta && q->mean && q->rstd && q->A && q->Bb && q->U && q->Y, "cvft_block_link_fwd: null operand, or head x != tail out");
    CVFT_CHECK_ARG(p->act == CVFT_ACT_GELU_ERF || p->act == CVFT_ACT_GELU_TANH, "cvft_block_link_fwd: act must be a GELU form");
    CVFT_CHECK_ARG(cvft_drop_rate_ok(q->p) && (q->p == 0.f || q->seed), "cvft_block_link_fwd: bad dropout args (p == 0 or 2^-16 <= p <= 1 - 2^-16, seed)");
    CVFT_CHECK_ARG(al16(p->x1) && al16(p->out) && al16(p->b1) && al16(p->b2) && al16(p->gamma) && al16(p->beta) && (!p->z || al16(p->z)) &&
                   al16(p->x0) && al16(p->bo) && al16(q->gamma) && al16(q->beta) && al16(q->A) && al16(q->Bb) && al16(q->Y) && (!q->bias || al16(q->bias)) &&
                   q->lda % 8 == 0 && q->ldb % 4 == 0 && q->ldu % 4 == 0 && q->ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(q->U) & 7) == 0 &&
                   (!q->y_out || al16(q->y_out)) && (!q->xd[0] || al16(q->xd[0])) && (!q->xd[1] || al16(q->xd[1])) && (!q->xd[2] || al16(q->xd[2])),
                   "cvft_block_link_fwd: operands must be 16-byte aligned (row pitches: ldo / lda %% 8, ldb / ldu / ldy %% 4)");
    TailFwd a;
    a.M = p->M; a.o = (const bf16_t*)p->o; a.ldo = p->ldo; a.x0 = (const bf16_t*)p->x0;
    a.Wst = (const bf16x8*)W_link; a.wave_frags = p->DI / 8 + p->F / 4 + q->N3 / 8; a.bo = p->bo; a.x1 = (bf16_t*)p->x1;
    a.gamma = p->gamma; a.beta = p->beta; a.eps = p->eps; a.b1 = p->b1; a.F = p->F; a.b2 = p->b2;
    a.z = (bf16_t*)p->z; a.mean = p->mean; a.rstd = p->rstd; a.out = (bf16_t*)p->out;
    QkvFwd h;
    h.M = q->M; h.x = (const bf16_t*)q->x; h.gamma = q->gamma; h.beta = q->beta; h.eps = q->eps; h.mean = q->mean; h.rstd = q->rstd;
    h.Wst = nullptr; h.wave_frags = q->N3 / 8; h.bias = q->bias; h.N3 = q->N3;
    h.A = (const bf16_t*)q->A; h.lda = q->lda; h.Bb = (const bf16_t*)q->Bb; h.ldb = q->ldb;
    h.alpha = q->alpha; h.p = q->p; h.seed = (const long long*)q->seed;
    for (int i = 0; i < 3; ++i) { h.sites[i] = q->sites[i]; h.xd[i] = (bf16_t*)q->xd[i]; }
    h.U = (bf16_t*)q->U; h.ldu = q->ldu; h.y_out = (bf16_t*)q->y_out; h.Y = (bf16_t*)q->Y; h.ldy = q->ldy;
    const bool erf = p->act == CVFT_ACT_GELU_ERF, drop = q->p > 0.f;
    hipStream_t st = (hipStream_t)stream;
    const int rc = erf ? (drop ? launch_link_fwd<CVFT_ACT_GELU_ERF, 2, true>(a, h, st) : launch_link_fwd<CVFT_ACT_GELU_ERF, 2, false>(a, h, st))
                       : (drop ? launch_link_fwd<CVFT_ACT_GELU_TANH, 2, true>(a, h, st) : launch_link_fwd<CVFT_ACT_GELU_TANH, 2, false>(a, h, st));
    if (rc) return rc;
    CVFT_LAUNCH_CHECK("cvft_block_link_fwd");
    return 0;
}

// the optional delta output of the tail backward (cvft.h): checked and copied into a kernel argument block
template <typename P>
static int tail_delta_args(const cvft_block_tail_bwd_args* p, const P*& ao, const P*& ao_lo, int& ldao, float*& delta, int& T, const char* who) {
    ao = nullptr; ao_lo = nullptr; ldao = 0; delta = nullptr; T = 1;
    if (!p->delta) return 0;
    if (!(p->dout && p->DI == 512 && p->attn_o && p->T > 0 && p->M % p->T == 0 && p->ldao % 4 == 0 && p->ldao >= p->DI &&
          (reinterpret_cast<uintptr_t>(p->attn_o) & 15) == 0 && (reinterpret_cast<uintptr_t>(p->attn_o_lo) & 15) == 0 && (p->lean == 0 || p->lean == 2))) {
        cvft_set_error("%s: delta needs dout with DI == 512, attn_o (16-byte aligned, ldao %% 4 == 0), M == B * T, form 0 or 2", who);
        return -1;
    }
    ao = (const P*)p->attn_o; ao_lo = (const P*)p->attn_o_lo; ldao = p->ldao; delta = p->delta; T = p->T;
    return 0;
}

template <int ACT, int CR, bool DROP>
static int launch_link_bwd(const QkvBwd& q, const TailBwd& a, hipStream_t st) {
    static int ready = 0;
    if (!ready) { if (bf_prepare(block_link_bwd_kernel<ACT, CR, DROP>, BF_LDS_TOTAL_DMA)) return -2; ready = 1; }
    hipLaunchKernelGGL((block_link_bwd_kernel<ACT, CR, DROP>), dim3((a.M + BF_ROWS - 1) / BF_ROWS), dim3(256), BF_LDS_TOTAL_DMA, st, q, a);
    return 0;
}

// The head backward of block i + 1 and the tail backward of block i on the same rows in one launch (include/cvft.h): both argument
// blocks mean what they mean for cvft_block_qkv_bwd / cvft_block_tail_bwd, with q->wide == 0, p->lean == 0 and p->dy == q->dx.
extern "C" int cvft_block_link_bwd(const cvft_block_qkv_bwd_args* q, const cvft_block_tail_bwd_args* p, const void* W_link, void* stream) {
    CVFT_CHECK_ARG(p && q && W_link && al16(W_link), "cvft_block_link_bwd: null / unaligned operand");
    CVFT_CHECK_ARG(p->M > 0 && q->M == p->M && p->F % 128 == 0 && p->F >= 256 && p->F <= 1024 && q->N3 == 1536,
                   "cvft_block_link_bwd: need equal M > 0, 256 <= F <= 1024 (F %% 128 == 0), 3N == 1536 (M=%d/%d F=%d)", p->M, q->M, p->F);
    CVFT_CHECK_ARG(p->lean == 0 && q->wide == 0 && p->dout && p->DI == 512 && p->lddo % 4 == 0 && al16(p->dout),
                   "cvft_block_link_bwd: the 32-row forms only (lean == 0, wide == 0), with the output projection's dgrad (DI == 512, dout)");
    CVFT_CHECK_ARG(q->dY && q->x && q->gamma && q->mean && q->rstd && q->At && q->Bbt && q->V && q->dx && p->x1 && p->dy == q->dx &&
                   p->gamma && p->mean && p->rstd && p->z && p->dx1, "cvft_block_link_bwd: null operand, or tail dy != head dx");
    CVFT_CHECK_ARG(p->act == CVFT_ACT_GELU_ERF || p->act == CVFT_ACT_GELU_TANH, "cvft_block_link_bwd: act must be a GELU form");
    CVFT_CHECK_ARG(cvft_drop_rate_ok(q->p) && (q->p == 0.f || q->seed), "cvft_block_link_bwd: bad dropout args (p == 0 or 2^-16 <= p <= 1 - 2^-16, seed)");
    CVFT_CHECK_ARG(al16(q->dY) && al16(q->x) && al16(q->gamma) && al16(q->Bbt) && al16(q->dx) && (!q->dres || al16(q->dres)) &&
                   q->lddy % 8 == 0 && q->ldbt % 8 == 0 && q->ldat % 4 == 0 && q->ldv % 4 == 0 && (reinterpret_cast<uintptr_t>(q->At) & 7) == 0 &&
                   (reinterpret_cast<uintptr_t>(q->V) & 7) == 0 && al16(p->x1) && al16(p->gamma) && al16(p->z) && al16(p->dx1),
                   "cvft_block_link_bwd: operands must be 16-byte aligned (row pitches: lddy / ldbt %% 8, ldat / ldv / lddo %% 4)");
    QkvBwd h;
    h.M = q->M; h.dY = (const bf16_t*)q->dY; h.lddy = q->lddy; h.dres = (const bf16_t*)q->dres; h.x = (const bf16_t*)q->x;
    h.gamma = q->gamma; h.mean = q->mean; h.rstd = q->rstd; h.Wst = (const bf16x8*)W_link;
    h.wave_frags = q->N3 / 8 + p->F / 4 + p->DI / 8; h.N3 = q->N3;
    h.At = (const bf16_t*)q->At; h.ldat = q->ldat; h.Bbt = (const bf16_t*)q->Bbt; h.ldbt = q->ldbt;
    h.alpha = q->alpha; h.p = q->p; h.seed = (const long long*)q->seed;
    for (int i = 0; i < 3; ++i) h.sites[i] = q->sites[i];
    h.V = (bf16_t*)q->V; h.ldv = q->ldv; h.dx = (bf16_t*)q->dx;
    TailBwd a;
    a.M = p->M; a.x1 = (const bf16_t*)p->x1; a.dy = (const bf16_t*)p->dy; a.gamma = p->gamma; a.mean = p->mean; a.rstd = p->rstd;
    a.z = (const bf16_t*)p->z; a.Wst = nullptr; a.wave_frags = p->F / 4 + p->DI / 8; a.F = p->F;
    a.dx1 = (bf16_t*)p->dx1; a.dout = (bf16_t*)p->dout; a.lddo = p->lddo;
    if (tail_delta_args(p, a.ao, a.ao_lo, a.ldao, a.delta, a.T, "cvft_block_link_bwd")) return -1;
    const bool erf = p->act == CVFT_ACT_GELU_ERF, drop = q->p > 0.f;
    hipStream_t st = (hipStream_t)stream;
    const int rc = erf ? (drop ? launch_link_bwd<CVFT_ACT_GELU_ERF, 2, true>(h, a, st) : launch_link_bwd<CVFT_ACT_GELU_ERF, 2, false>(h, a, st))
                       : (drop ? launch_link_bwd<CVFT_ACT_GELU_TANH, 2, true>(h, a, st) : launch_link_bwd<CVFT_ACT_GELU_TANH, 2, false>(h, a, st));
    if (rc) return rc;
    CVFT_LAUNCH_CHECK("cvft_block_link_bwd");
    return 0;
}

extern "C" int cvft_block_tail_bwd(const cvft_block_tail_bwd_args* p, void* stream) {
    CVFT_CHECK_ARG(p && p->M > 0 && p->F > 0 && p->F % 128 == 0 && p->F <= BF_MAX_F, "cvft_block_tail_bwd: need M > 0, F %% 128 == 0, F <= 2048");
    CVFT_CHECK_ARG(p->x1 && p->dy && p->gamma && p->mean && p->rstd && p->z && p->W_bwd && p->dx1, "cvft_block_tail_bwd: null operand");
    CVFT_CHECK_ARG(p->act == CVFT_ACT_GELU_ERF || p->act == CVFT_ACT_GELU_TANH, "cvft_block_tail_bwd: act must be a GELU form");
    const int DI = p->dout ? p->DI : 0;
    if (p->dout) CVFT_CHECK_ARG((DI == 256 || DI == 512) && p->lddo % 4 == 0 && al16(p->dout), "cvft_block_tail_bwd: output-projection dgrad needs DI in {256, 512}, aligned dout");
    CVFT_CHECK_ARG(al16(p->x1) && al16(p->dy) && al16(p->gamma) && al16(p->z) && al16(p->W_bwd) && al16(p->dx1), "cvft_block_tail_bwd: operands must be 16-byte aligned");
    if (p->lean) {
        CVFT_CHECK_ARG(p->lean == 1 || (p->lean == 2 && p->F >= 256), "cvft_block_tail_bwd: lean must be 0, 1 or 2 (2: F >= 256)");
        {
            const bf16_t* t0; const bf16_t* t1; int t2, t4; float* t3;      // (validates the optional delta request: forms 0 and 2 only)
            if (tail_delta_args(p, t0, t1, t2, t3, t4, "cvft_block_tail_bwd")) return -1;
            CVFT_CHECK_ARG(!p->delta || p->ldao % 8 == 0, "cvft_block_tail_bwd: the wide form reads attn_o in 16-byte pieces (ldao %% 8 == 0)");
        }
        CVFT_CHECK_ARG(p->lean == 1 || !p->dout || p->lddo % 8 == 0, "cvft_block_tail_bwd: the wide form stores dout in 16-byte pieces (lddo %% 8 == 0)");
        const int rc = p->lean == 2 ? block_tail_wide_bwd_launch(p, DI, stream) : block_tail_lean_bwd_launch(p, DI, stream);
        if (rc) return rc;
        CVFT_LAUNCH_CHECK("cvft_block_tail_bwd (lean)");
        return 0;
    }
    TailBwd a;
    a.M = p->M; a.x1 = (const bf16_t*)p->x1; a.dy = (const bf16_t*)p->dy; a.gamma = p->gamma; a.mean = p->mean; a.rstd = p->rstd;
    a.z = (const bf16_t*)p->z; a.Wst = (const bf16x8*)p->W_bwd; a.wave_frags = p->F / 4 + p->DI / 8; a.F = p->F;
    a.dx1 = (bf16_t*)p->dx1; a.dout = (bf16_t*)p->dout; a.lddo = p->lddo;
    if (tail_delta_args(p, a.ao, a.ao_lo, a.ldao, a.delta, a.T, "cvft_block_tail_bwd")) return -1;
    const bool erf = p->act == CVFT_ACT_GELU_ERF;
    int rc;
    hipStream_t st = (hipStream_t)stream;
    if (DI == 0) rc = erf ? launch_bwd<CVFT_ACT_GELU_ERF, 0>(a, st) : launch_bwd<CVFT_ACT_GELU_TANH, 0>(a, st);
    else if (DI == 256) rc = erf ? launch_bwd<CVFT_ACT_GELU_ERF, 1>(a, st) : launch_bwd<CVFT_ACT_GELU_TANH, 1>(a, st);
    else rc = erf ? launch_bwd<CVFT_ACT_GELU_ERF, 2>(a, st) : launch_bwd<CVFT_ACT_GELU_TANH, 2>(a, st);
    if (rc) return rc;
    CVFT_LAUNCH_CHECK("cvft_block_tail_bwd");
    return 0;
}
