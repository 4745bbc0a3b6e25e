// block_wide8.hip -- the block-tail FORWARD with 64 rows and EIGHT waves per workgroup (two per SIMD, <= 256 registers each).
//
// block_wide.hip's four-wave forward is issue-bound: one wave per SIMD pays for every VALU instruction of the activation on top of
// its MFMA issue time (5.3 k ticks per round where the weight stream would allow 2.6 k).  Two waves per SIMD
// fill each other's issue gaps (tools/ub/mfma_valu.hip), as in block_qkv_wide.hip -- but 256 registers hold neither a 32-fragment
// ring nor a row tile's B fragments, so: a ring of 16 fragments per wave (= one group of the stream; 8 x 16 KB in flight per CU as
// before), B fragments from LDS one step ahead, ONE accumulator pair for the first product (the round's pre-activations wait as
// packed bf16 -- what the reference's Linear hands to GELU under autocast), activation results streamed out in 8-byte pieces.
// Wave w (0..7) owns output features 32 w .. 32 w + 31 of every link, both row tiles, and hidden tile 8 r + w of round r (rounds of
// 256 hidden units); the second product lags one round (block_wide.hip) and the stream is packed in that order (blockpack.py,
// W_fwd_wide8).  z (= gelu'(z), block_common.h bf_gelu2) has the order every other form uses: the four-wave 64-row backward reads it.
#include "block_common.h"

#define B8_ROWS 64
#define B8_THREADS 512
#define B8_RING 16
#define B8_MAX_F 1024
#define B8_HT_TILE 2560                         // one hidden tile: [32 rows][80 B]
// LDS carve
#define B8_HT 0                                 // 2 buffers x 8 hidden tiles x 2 row tiles (80 KB); the o tiles 2 x [32][512] first
#define B8_XT 81920                             // x0 -> x1 -> out tiles 2 x [32][256]
#define B8_YT (B8_XT + 32768)                   // LN(x1) tiles
#define B8_STAT (B8_YT + 32768)                 // 2 quantities x 2 row tiles x 8 waves x 32 rows
#define B8_BIAS (B8_STAT + 2 * 2 * 8 * 32 * 4)
#define B8_PAR (B8_BIAS + 4 * B8_MAX_F)
#define B8_TOTAL (B8_PAR + 4 * 4 * BF_D)        // 159 744 B

__device__ __forceinline__ bf16x8 b8_frag256(const char* tile, int m, int h, int ks) {
    return *reinterpret_cast<const bf16x8*>(tile + m * 512 + (((2 * ks + h) ^ (m & 15)) << 4));
}
__device__ __forceinline__ bf16x8 b8_frag512(const char* tile, int m, int h, int ks) {
    return *reinterpret_cast<const bf16x8*>(tile + m * 1024 + (((2 * ks + h) ^ (m & 15)) << 4));
}
__device__ __forceinline__ bf16x8 b8_get_frag(const char* tile, int m, int h, int s) {
    return *reinterpret_cast<const bf16x8*>(tile + m * 80 + (16 * s + 8 * h) * 2);
}
struct B8Touch { unsigned v[4]; };

struct Wide8Fwd {
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

// one group of the stream: acc[t] += ring[k] . (B fragment of row tile t, step k, read one step ahead from LDS by FR);  RELOAD: slot k
// re-requested for the next group (nx = this group's first fragment + 16);  SIDE(k): other work fenced behind step k's MFMAs
#define B8_PROD(RELOAD, FR, ACC, SIDE)                                                              \
    do {                                                                                            \
        bf16x8 bq_[2][2];                                                                           \
        _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_) bq_[0][t_] = FR(t_, 0);                    \
        _Pragma("unroll") for (int k_ = 0; k_ < B8_RING; ++k_) {                                    \
            if (k_ + 1 < B8_RING) {                                                                 \
                _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_) bq_[(k_ + 1) & 1][t_] = FR(t_, k_ + 1); \
            }                                                                                       \
            _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_) ACC[t_] = mfma32(ring[k_], bq_[k_ & 1][t_], ACC[t_]); \
            if (RELOAD) ring[k_] = nx[k_ * 64];                                                     \
            SIDE(k_);                                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                      \
        }                                                                                           \
        nx += B8_RING * 64;                                                                         \
    } while (0)

// AR = DI / 256.  Stream per wave w of 8 (blockpack.py, W_fwd_wide8), groups of 16 fragments: AR groups of the output projection
// (feature tile w, k-steps in order); then, with G1(r) = W1 of hidden tile 8 r + w ([ks]) and G2(r) = W2 of feature tile w over round
// r's 256 hidden units ([k']):  G1(0), G1(1), { G1(r + 1), G2(r - 1) : r = 1 .. nr - 2 }, G2(nr - 2), G2(nr - 1)   (nr = F / 256 >= 2)
template <int ACT, int AR>
__global__ __launch_bounds__(B8_THREADS, 1) void block_tail_wide8_fwd_kernel(Wide8Fwd a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * B8_ROWS;

    BF_STAMP(0);
    // ---- requests in the order their data is needed: o tiles, the ring, x0 / parameters, touches
    constexpr int NO = AR > 0 ? 4 * AR : 1;            // 16-byte chunks of o per thread (64 rows x DI / 8 chunks)
    bf16x8 ov[NO];
    if (AR > 0) {
        constexpr int CPR = AR > 0 ? 32 * AR : 1;
#pragma unroll
        for (int i = 0; i < NO; ++i) {
            const int q = i * B8_THREADS + threadIdx.x;
            const int r = q / CPR, ch = q % CPR;
            ov[i] = *reinterpret_cast<const bf16x8*>(a.o + (size_t)min(m0 + r, a.M - 1) * a.ldo + 8 * ch);
        }
    }
    const bf16x8* nx = a.Wst + (size_t)wave * a.wave_frags * 64 + lane;
    bf16x8 ring[B8_RING];
#pragma unroll
    for (int i = 0; i < B8_RING; ++i) ring[i] = nx[i * 64];
    nx += B8_RING * 64;
    bf16x8 xv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = i * B8_THREADS + threadIdx.x;
        const int r = q >> 5, ch = q & 31;
        xv[i] = *reinterpret_cast<const bf16x8*>((AR > 0 ? a.x0 : a.x1) + (size_t)min(m0 + r, a.M - 1) * BF_D + 8 * ch);
    }
    f32x4 pv = {0.f, 0.f, 0.f, 0.f}, b1v = {0.f, 0.f, 0.f, 0.f};
    if (wave < 4) {
        const float* srcp = wave == 0 ? a.bo : wave == 1 ? a.gamma : wave == 2 ? a.beta : a.b2;
        if (srcp != nullptr) pv = reinterpret_cast<const f32x4*>(srcp)[lane];
    }
    if ((int)threadIdx.x < a.F / 4) b1v = reinterpret_cast<const f32x4*>(a.b1)[threadIdx.x];
    B8Touch touched;
    if (BF_TOUCH) {
        const int lines = 8 * a.wave_frags;            // one 128-byte line in eight, an eighth of them per workgroup
        const char* base = reinterpret_cast<const char*>(a.Wst) + (size_t)((blockIdx.x >> 3) & 7) * lines * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            touched.v[i] = *reinterpret_cast<const unsigned*>(base + (size_t)min(i * B8_THREADS + (int)threadIdx.x, lines - 1) * 128);
    }
    if (AR > 0) {
        constexpr int CPR = AR > 0 ? 32 * AR : 1;
#pragma unroll
        for (int i = 0; i < NO; ++i) {
            const int q = i * B8_THREADS + threadIdx.x;
            const int r = q / CPR, ch = q % CPR;
            *reinterpret_cast<bf16x8*>(smem + B8_HT + (r >> 5) * 32768 + (r & 31) * 1024 + ((ch ^ (r & 15)) << 4)) = ov[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = i * B8_THREADS + threadIdx.x;
        const int r = q >> 5, ch = q & 31;
        *reinterpret_cast<bf16x8*>(smem + B8_XT + (r >> 5) * 16384 + (r & 31) * 512 + ((ch ^ (r & 15)) << 4)) = xv[i];
    }
    if (wave < 4) reinterpret_cast<f32x4*>(smem + B8_PAR)[wave * 64 + lane] = pv;
    if ((int)threadIdx.x < a.F / 4) reinterpret_cast<f32x4*>(smem + B8_BIAS)[threadIdx.x] = b1v;
    const float* pbo = reinterpret_cast<const float*>(smem + B8_PAR), *pgam = pbo + BF_D, *pbet = pbo + 2 * BF_D, *pb2 = pbo + 3 * BF_D;
    __syncthreads();                                   // o, x0 tiles and parameters are in LDS
    BF_STAMP(1);

#define B8_NONE(k)
#define B8_OFRAG(t, k) b8_frag512(smem + B8_HT + (t) * 32768, m, h, 16 * pass + (k))
#define B8_YFRAG(t, k) b8_frag256(smem + B8_YT + (t) * 16384, m, h, (k))
    {
        bf16x4 xb[2][4];                               // this wave's 32 features of x1, both row tiles (accumulator layout)
        if (AR > 0) {
            // ---- x1 = x0 + o Wo^T + bo for this wave's 32 features
            f32x16 acc[2] = {zero16(), zero16()};
#pragma unroll
            for (int pass = 0; pass < AR; ++pass) B8_PROD(true, B8_OFRAG, acc, B8_NONE);
            BF_STAMP(2);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = 32 * wave + 8 * g + 4 * h;
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(pbo + c);
                    char* px = smem + B8_XT + t * 16384 + bf_tile_off(m, c);
                    const bf16x4 x0v = *reinterpret_cast<const bf16x4*>(px);
#pragma unroll
                    for (int i = 0; i < 4; ++i) xb[t][g][i] = (bf16_t)(acc[t][4 * g + i] + bb[i] + (float)x0v[i]);
                    *reinterpret_cast<bf16x4*>(px) = xb[t][g];
                }
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    xb[t][g] = *reinterpret_cast<const bf16x4*>(smem + B8_XT + t * 16384 + bf_tile_off(m, 32 * wave + 8 * g + 4 * h));
        }
        BF_STAMP(3);
        // ---- LayerNorm -> y tiles.  Statistics with ONE exchange: every wave's mean and centred sum of squares over its own 32
        // features, combined as a parallel variance: M2 = sum M2_w + 32 sum (mean_w - mean)^2
        float* st = reinterpret_cast<float*>(smem + B8_STAT);
        float mw[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float sw = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) sw += (float)xb[t][g][i];
            sw += __shfl_xor(sw, 32, 64);
            mw[t] = sw * (1.f / 32.f);
            float qw = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float d = (float)xb[t][g][i] - mw[t]; qw += d * d; }
            qw += __shfl_xor(qw, 32, 64);
            if (lane < 32) { st[(0 * 2 + t) * 256 + wave * 32 + lane] = mw[t]; st[(1 * 2 + t) * 256 + wave * 32 + lane] = qw; }
        }
        __syncthreads();                               // (also: every wave's x1 is in the x tiles, nobody reads the o tiles any more)
        if (AR > 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = i * B8_THREADS + threadIdx.x;
                const int r = q >> 5, ch = q & 31;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + B8_XT + (r >> 5) * 16384 + (r & 31) * 512 + ((ch ^ (r & 15)) << 4));
                if (m0 + r < a.M) *reinterpret_cast<bf16x8*>(a.x1 + (size_t)(m0 + r) * BF_D + 8 * ch) = v;
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float mu = 0.f, m2 = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) { mu += st[(0 * 2 + t) * 256 + w * 32 + m]; m2 += st[(1 * 2 + t) * 256 + w * 32 + m]; }
            mu *= 0.125f;
#pragma unroll
            for (int w = 0; w < 8; ++w) { const float d = st[(0 * 2 + t) * 256 + w * 32 + m] - mu; m2 = fmaf(32.f * d, d, m2); }
            const float rstd = rsqrtf(m2 * (1.f / BF_D) + a.eps);
            const int rowt = m0 + 32 * t + m;
            if (wave == 0 && lane < 32 && rowt < a.M) { a.mean[rowt] = mu; a.rstd[rowt] = rstd; }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 32 * wave + 8 * g + 4 * h;
                const f32x4 gg = *reinterpret_cast<const f32x4*>(pgam + c);
                const f32x4 be = *reinterpret_cast<const f32x4*>(pbet + c);
                bf16x4 y;
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] = (bf16_t)(((float)xb[t][g][i] - mu) * rstd * gg[i] + be[i]);
                *reinterpret_cast<bf16x4*>(smem + B8_YT + t * 16384 + bf_tile_off(m, c)) = y;
            }
        }
    }
    __syncthreads();
    BF_STAMP(4);

    // ---- feed-forward in rounds of eight hidden tiles (wave w: tile 8 r + w); this wave's 32 output features in acc2
    const int nr = a.F / 256;
    const float* b1s = reinterpret_cast<const float*>(smem + B8_BIAS);
    f32x16 acc2[2] = {zero16(), zero16()}, acc1[2];
    bf16x8 zb[2][2];                                    // the round's pre-activations, bf16
    // z of whole 64-row groups (the entry point checks that z is given and holds them): gelu'(z), 8-byte pieces
    bf16x4* zp = reinterpret_cast<bf16x4*>(a.z) + ((size_t)(blockIdx.x * 2 * (a.F / 32) + wave) * 64 + lane) * 4;
    const size_t ztile = (size_t)(a.F / 32) * 256;      // bf16x4 units between the two row tiles of a group
    int rc = 0;
    bf16x4 zcur, hcur;
    // activation of element e of round rc (e >> 4 = row tile, e & 15 = accumulator register), called in increasing e: the derivative goes to
    // z and the activated value into the round's exchange buffer (rc & 1: free since the barrier that ended round rc - 1) per 8-byte piece
#define B8_ACT(e)                                                                                                   \
    do {                                                                                                            \
        const int t_a = (e) >> 4, i_a = (e) & 15;                                                                   \
        float g_a, dg_a;                                                                                            \
        bf_gelu2<ACT>((float)zb[t_a][i_a >> 3][i_a & 7], g_a, dg_a);                                                \
        zcur[i_a & 3] = (bf16_t)dg_a;                                                                               \
        hcur[i_a & 3] = (bf16_t)g_a;                                                                                \
        if ((i_a & 3) == 3) {                                                                                       \
            *reinterpret_cast<bf16x4*>(smem + B8_HT + ((rc & 1) * 16 + wave * 2 + t_a) * B8_HT_TILE + m * 80 + (8 * (i_a >> 2) + 4 * h) * 2) = hcur; \
            zp[t_a * ztile + (size_t)rc * 8 * 256 + (i_a >> 2)] = zcur;                                             \
        }                                                                                                           \
    } while (0)
#define B8_ACT2(k) do { B8_ACT(2 * (k)); B8_ACT(2 * (k) + 1); } while (0)       /* a whole round beside one group */
#define B8_ACT_LO(k) B8_ACT(k)                                                  /* row tile 0 beside G1 */
#define B8_ACT_HI(k) B8_ACT(16 + (k))                                           /* row tile 1 beside G2 */
#define B8_HFRAG(t, k) b8_get_frag(htr + (((k) >> 1) * 2 + (t)) * B8_HT_TILE, m, h, (k) & 1)
#define B8_FIRST(r, SIDE)                                                                           \
    do {                                                                                            \
        _Pragma("unroll") for (int t = 0; t < 2; ++t) acc1[t] = bf_bias_init(b1s, 8 * (r) + wave, h); \
        B8_PROD(true, B8_YFRAG, acc1, SIDE);                                                        \
    } while (0)
#define B8_KEEP()                                                                                   \
    do {                                                                                            \
        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                               \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) zb[t][i >> 3][i & 7] = (bf16_t)acc1[t][i]; \
    } while (0)
    B8_FIRST(0, B8_NONE);                               // G1(0)
    B8_KEEP();
    BF_STAMP(5);
    B8_FIRST(1, B8_ACT2);                               // round 0: its activation beside G1(1)
    __syncthreads();
    B8_KEEP();
    BF_STAMP(8);
    for (int r = 1; r + 1 < nr; ++r) {
        rc = r;
        B8_FIRST(r + 1, B8_ACT_LO);                     // round r: row tile 0's activation beside G1(r + 1) ...
        const char* htr = smem + B8_HT + ((r - 1) & 1) * 16 * B8_HT_TILE;
        BF_STAMP(9 + 2 * (r - 1));
        B8_PROD(true, B8_HFRAG, acc2, B8_ACT_HI);       // ... row tile 1's beside G2(r - 1)
        __syncthreads();
        B8_KEEP();
        BF_STAMP(10 + 2 * (r - 1));
    }
    BF_STAMP(6);
    rc = nr - 1;
    {
        const char* htr = smem + B8_HT + ((nr - 2) & 1) * 16 * B8_HT_TILE;
        B8_PROD(true, B8_HFRAG, acc2, B8_ACT2);         // last round: its activation beside G2(nr - 2)
    }
    __syncthreads();
    {
        const char* htr = smem + B8_HT + ((nr - 1) & 1) * 16 * B8_HT_TILE;
        B8_PROD(false, B8_HFRAG, acc2, B8_NONE);        // G2(nr - 1)
    }
    BF_STAMP(7);
    // ---- out = x1 + h W2^T + b2: into the x tiles (each lane over its own x1 values), then row-major to global
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 32 * wave + 8 * g + 4 * h;
            const f32x4 bb = *reinterpret_cast<const f32x4*>(pb2 + c);
            char* px = smem + B8_XT + t * 16384 + bf_tile_off(m, c);
            const bf16x4 x1r = *reinterpret_cast<const bf16x4*>(px);
            bf16x4 yo;
#pragma unroll
            for (int i = 0; i < 4; ++i) yo[i] = (bf16_t)(acc2[t][4 * g + i] + bb[i] + (float)x1r[i]);
            *reinterpret_cast<bf16x4*>(px) = yo;
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = i * B8_THREADS + threadIdx.x;
        const int r = q >> 5, ch = q & 31;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + B8_XT + (r >> 5) * 16384 + (r & 31) * 512 + ((ch ^ (r & 15)) << 4));
        if (m0 + r < a.M) *reinterpret_cast<bf16x8*>(a.out + (size_t)(m0 + r) * BF_D + 8 * ch) = v;
    }
    BF_STAMP(26);
    if (BF_TOUCH && (touched.v[0] | touched.v[1] | touched.v[2] | touched.v[3]) == 0x7fc07fc1u && a.M < 0) a.mean[0] = 0.f;   // (keeps the touches alive)
}

template <int ACT, int AR>
static int launch_wide8_fwd(const Wide8Fwd& a, hipStream_t st) {
    static int ready = 0;
    if (!ready) {
        if (hipFuncSetAttribute((const void*)block_tail_wide8_fwd_kernel<ACT, AR>, hipFuncAttributeMaxDynamicSharedMemorySize, B8_TOTAL) != hipSuccess) {
            cvft_set_error("block_wide8: cannot reserve %d bytes of LDS", B8_TOTAL);
            return -2;
        }
        ready = 1;
    }
    hipLaunchKernelGGL((block_tail_wide8_fwd_kernel<ACT, AR>), dim3((a.M + B8_ROWS - 1) / B8_ROWS), dim3(B8_THREADS), B8_TOTAL, st, a);
    return 0;
}

// called by cvft_block_tail_fwd (block_fused.hip) when args.lean == 4; arguments are already checked there
// (F % 256 == 0, 512 <= F <= 1024, z given)
int block_tail_wide8_fwd_launch(const cvft_block_tail_args* p, int DI, void* stream) {
    Wide8Fwd a;
    a.M = p->M; a.o = (const bf16_t*)p->o; a.ldo = p->ldo; a.x0 = (const bf16_t*)p->x0;
    a.Wst = (const bf16x8*)p->W_fwd; a.wave_frags = DI / 16 + p->F / 8; a.bo = p->bo; a.x1 = (bf16_t*)p->x1;
    a.gamma = p->gamma; a.beta = p->beta; a.eps = p->eps; a.b1 = p->b1; a.F = p->F; a.b2 = p->b2;
    a.z = (bf16_t*)p->z; a.mean = p->mean; a.rstd = p->rstd; a.out = (bf16_t*)p->out;
    const bool erf = p->act == CVFT_ACT_GELU_ERF;
    hipStream_t st = (hipStream_t)stream;
    if (DI == 0) return erf ? launch_wide8_fwd<CVFT_ACT_GELU_ERF, 0>(a, st) : launch_wide8_fwd<CVFT_ACT_GELU_TANH, 0>(a, st);
    if (DI == 256) return erf ? launch_wide8_fwd<CVFT_ACT_GELU_ERF, 1>(a, st) : launch_wide8_fwd<CVFT_ACT_GELU_TANH, 1>(a, st);
    return erf ? launch_wide8_fwd<CVFT_ACT_GELU_ERF, 2>(a, st) : launch_wide8_fwd<CVFT_ACT_GELU_TANH, 2>(a, st);
}
