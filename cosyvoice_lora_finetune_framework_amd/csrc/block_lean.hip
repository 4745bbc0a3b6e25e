// block_lean.hip -- the block-tail chain kernels (block_fused.hip) in a form that SHARES its CU: <= 256 registers per lane and
// ~82 KB of LDS, so a workgroup fits beside one 128x128 GEMM block (8 waves x <= 128 registers, 64 KB) of another chain instead
// of waiting for a completely empty CU ("Concurrency").
//
// Same math, same weight-stream skeleton (ring of 32 fragments, block_fused.hip), different ownership: every wave owns 64 OUTPUT
// features of each link (two 32-feature tiles = 32 accumulator registers instead of 128) and one hidden tile per round of four;
// the round's four activated hidden tiles are exchanged through a double-buffered LDS tile (one barrier per round), the
// activations (o, LN(x1), dy, dx1) are read as B fragments from LDS tiles instead of living in registers, and there is no
// cross-wave partial-sum exchange at all.
#include "block_common.h"

// LDS carve of the lean kernels
#define BL_OT 0                            // [32][512] bf16 o tile (forward) -- 16-byte chunks XOR-swizzled by (row & 15): 32 KB
#define BL_YT 32768                        // [32][256] bf16: LN(x1) (forward) / dy, then dx1 (backward): 16 KB
#define BL_HT (BL_YT + 16384)              // 2 buffers x 4 tiles x [32 rows][80 B]: the round's hidden tiles, 20 KB
#define BL_HT_TILE 2560
#define BL_STAT (BL_HT + 2 * 4 * BL_HT_TILE)
#define BL_BIAS (BL_STAT + 3 * 4 * 32 * 4)
#define BL_PAR (BL_BIAS + 4 * BF_MAX_F)
#define BL_TOTAL (BL_PAR + 4 * 4 * BF_D)

__device__ __forceinline__ float bl_rowsum(char* smem, int slot, int wave, int lane, float partial) {
    float* st = reinterpret_cast<float*>(smem + BL_STAT) + slot * 128;
    partial += __shfl_xor(partial, 32, 64);
    if (lane < 32) st[wave * 32 + lane] = partial;
    __syncthreads();
    const int m = lane & 31;
    return st[m] + st[32 + m] + st[64 + m] + st[96 + m];
}
// B fragment of k-step ks from a [32][256] swizzled tile / from the [32][512] o tile
__device__ __forceinline__ bf16x8 bl_frag256(const char* tile, int m, int h, int ks) {
    return *reinterpret_cast<const bf16x8*>(tile + m * 512 + (((2 * ks + h) ^ (m & 15)) << 4));
}
__device__ __forceinline__ bf16x8 bl_frag512(const char* tile, int m, int h, int ks) {
    return *reinterpret_cast<const bf16x8*>(tile + m * 1024 + (((2 * ks + h) ^ (m & 15)) << 4));
}
// this wave's activated hidden tile (accumulator layout: lane = row, register q = feature (q&3) + 8 (q>>2) + 4 h) -> LDS tile
__device__ __forceinline__ void bl_put_tile(char* tile, int m, int h, const bf16x8 (&hb)[2]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const bf16x8& s = hb[g >> 1];
        const int o = (g & 1) * 4;
        *reinterpret_cast<bf16x4*>(tile + m * 80 + (8 * g + 4 * h) * 2) = bf16x4{s[o], s[o + 1], s[o + 2], s[o + 3]};
    }
}
__device__ __forceinline__ bf16x8 bl_get_frag(const char* tile, int m, int h, int s) {
    return *reinterpret_cast<const bf16x8*>(tile + m * 80 + (16 * s + 8 * h) * 2);
}

// second product of a round: acc[c2] += (this wave's two feature tiles of the streamed weight, ring slots S0 .. S0+15 in order
// [k'][c2]) . (the round's four hidden tiles, read as B fragments from the exchange buffer);  RELOAD: re-request the 16 slots for
// stream positions + 32 (nx = address of this group's first fragment + 32)
template <int S0, bool RELOAD>
__device__ __forceinline__ void bl_second(bf16x8 (&ring)[BF_RING], const bf16x8* nx, const char* ht, int m, int h, f32x16 (&acc)[2]) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const bf16x8 hf = bl_get_frag(ht + (k >> 1) * BL_HT_TILE, m, h, k & 1);
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            const int j = 2 * k + c2;
            acc[c2] = mfma32(ring[S0 + j], hf, acc[c2]);
            if (RELOAD) ring[S0 + j] = nx[j * 64];
        }
    }
}

struct LeanFwd {
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

// AR = DI / 256 (0 = no output projection, x1 is the input).  Stream per wave w (blockpack.py, "lean"): projection fragments of
// its feature tiles 2w, 2w+1 in order [ks][c2] (DI / 8 of them); then W1 of hidden tile w; then per round r < F/128:
// [W1 of hidden tile 4 (r+1) + w], W2 of its two feature tiles over the round's 128 hidden units in order [k'][c2].
template <int ACT, int AR>
__global__ __launch_bounds__(256, 2) void block_tail_lean_fwd_kernel(LeanFwd a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * BF_ROWS;
    const int row = min(m0 + m, a.M - 1);
    const bool rvalid = m0 + m < a.M;

    // ---- loads in the order they are needed: the o tile (-> LDS), this wave's residual features, parameters (-> LDS), the ring
    if (AR > 0) {
        constexpr int CPR = AR > 0 ? 32 * AR : 1;      // 16-byte chunks per row
#pragma unroll
        for (int i = 0; i < 4 * AR; ++i) {             // 32 rows x DI / 8 chunks of 16 B, 256 threads
            const int q = i * 256 + threadIdx.x;
            const int r = q / CPR, ch = q % CPR;
            const int rr = min(m0 + r, a.M - 1);
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(a.o + (size_t)rr * a.ldo + 8 * ch);
            *reinterpret_cast<bf16x8*>(smem + BL_OT + r * 1024 + ((ch ^ (r & 15)) << 4)) = v;
        }
    }
    bf16x4 xb[2][4];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            xb[c2][g] = *reinterpret_cast<const bf16x4*>((AR > 0 ? a.x0 : a.x1) + (size_t)row * BF_D + 64 * wave + 32 * c2 + 8 * g + 4 * h);
    {
        f32x4* par = reinterpret_cast<f32x4*>(smem + BL_PAR);
        const float* srcp = wave == 0 ? a.bo : wave == 1 ? a.gamma : wave == 2 ? a.beta : a.b2;
        if (srcp != nullptr) par[wave * 64 + lane] = reinterpret_cast<const f32x4*>(srcp)[lane];
        for (int k = threadIdx.x; k < a.F / 4; k += 256)
            reinterpret_cast<f32x4*>(smem + BL_BIAS)[k] = reinterpret_cast<const f32x4*>(a.b1)[k];
    }
    const bf16x8* nx = a.Wst + (size_t)wave * a.wave_frags * 64 + lane;
    bf16x8 ring[BF_RING];
#pragma unroll
    for (int i = 0; i < BF_RING; ++i) ring[i] = nx[i * 64];
    nx += BF_RING * 64;
    BfTouch touched;
    if (BF_TOUCH) touched = bf_touch_stream(a.Wst, 4 * a.wave_frags);
    const float* pbo = reinterpret_cast<const float*>(smem + BL_PAR), *pgam = pbo + BF_D, *pbet = pbo + 2 * BF_D, *pb2 = pbo + 3 * BF_D;
    __syncthreads();                                   // o tile and parameters are in LDS

    if (AR > 0) {
        // ---- x1 = x0 + o Wo^T + bo for this wave's 64 features: no k split, no exchange
        f32x16 acc[2] = {zero16(), zero16()};
#pragma unroll
        for (int rr = 0; rr < AR; ++rr) {
#pragma unroll
            for (int j = 0; j < BF_RING; ++j) {
                const bf16x8 of = bl_frag512(smem + BL_OT, m, h, 16 * rr + (j >> 1));
                acc[j & 1] = mfma32(ring[j], of, acc[j & 1]);
                ring[j] = nx[j * 64];
            }
            nx += BF_RING * 64;
        }
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 64 * wave + 32 * c2 + 8 * g + 4 * h;
                const f32x4 bb = *reinterpret_cast<const f32x4*>(pbo + c);
#pragma unroll
                for (int i = 0; i < 4; ++i) xb[c2][g][i] = (bf16_t)(acc[c2][4 * g + i] + bb[i] + (float)xb[c2][g][i]);
                if (rvalid) *reinterpret_cast<bf16x4*>(a.x1 + (size_t)row * BF_D + c) = xb[c2][g];
            }
    }
    // ---- LayerNorm -> y tile
    float s = 0.f;
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) s += (float)xb[c2][g][i];
    const float mean = bl_rowsum(smem, 0, wave, lane, s) * (1.f / BF_D);
    s = 0.f;
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float d = (float)xb[c2][g][i] - mean; s += d * d; }
    const float rstd = rsqrtf(bl_rowsum(smem, 1, wave, lane, s) * (1.f / BF_D) + a.eps);
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
            *reinterpret_cast<bf16x4*>(smem + BL_YT + bf_tile_off(m, c)) = y;
        }
    __syncthreads();

    // ---- feed-forward in rounds of four hidden tiles (wave w: tile 4 r + w); this wave's 64 output features in acc2
    const int nr = a.F / 128;
    const float* b1s = reinterpret_cast<const float*>(smem + BL_BIAS);
    f32x16 acc2[2] = {zero16(), zero16()};
    f32x16 acc1 = bf_bias_init(b1s, wave, h);
#pragma unroll
    for (int ks = 0; ks < BF_KS; ++ks) {
        acc1 = mfma32(ring[ks], bl_frag256(smem + BL_YT, m, h, ks), acc1);
        ring[ks] = nx[ks * 64];
    }
    nx += BF_KS * 64;
    bf16x8* zp = a.z == nullptr ? nullptr : reinterpret_cast<bf16x8*>(a.z) + ((size_t)(blockIdx.x * (a.F / 32) + wave) * 64 + lane) * 2;
    // (no software pipelining across the activation here: with two waves per SIMD -- this kernel's partner workgroup or a GEMM block
    // of another chain -- the other wave's MFMAs cover this wave's VALU, and 16 registers of a second accumulator do not fit)
    for (int r = 0; r + 1 < nr; ++r) {
        bf16x8 hb[2], zs[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) {                  // (the workspace takes gelu'(z): block_common.h, bf_gelu2)
            float g, dg;
            bf_gelu2<ACT>(acc1[i], g, dg);
            zs[i >> 3][i & 7] = (bf16_t)dg;
            hb[i >> 3][i & 7] = (bf16_t)g;
        }
        acc1 = bf_bias_init(b1s, 4 * (r + 1) + wave, h);
#pragma unroll
        for (int ks = 0; ks < BF_KS; ++ks) {
            acc1 = mfma32(ring[16 + ks], bl_frag256(smem + BL_YT, m, h, ks), acc1);
            ring[16 + ks] = nx[ks * 64];
        }
        if (zp != nullptr) {
            zp[(size_t)r * 4 * 128] = zs[0];
            zp[(size_t)r * 4 * 128 + 1] = zs[1];
        }
        char* ht = smem + BL_HT + (r & 1) * 4 * BL_HT_TILE;
        bl_put_tile(ht + wave * BL_HT_TILE, m, h, hb);
        __syncthreads();
        bl_second<0, true>(ring, nx + 16 * 64, ht, m, h, acc2);       // (the W1 group went through slots 16..31)
        nx += BF_RING * 64;
    }
    {                                                  // last round: its W2 group sits in slots 16..31
        bf16x8 hb[2], zs[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) {                  // (the workspace takes gelu'(z): block_common.h, bf_gelu2)
            float g, dg;
            bf_gelu2<ACT>(acc1[i], g, dg);
            zs[i >> 3][i & 7] = (bf16_t)dg;
            hb[i >> 3][i & 7] = (bf16_t)g;
        }
        if (zp != nullptr) {
            zp[(size_t)(nr - 1) * 4 * 128] = zs[0];
            zp[(size_t)(nr - 1) * 4 * 128 + 1] = zs[1];
        }
        char* ht = smem + BL_HT + ((nr - 1) & 1) * 4 * BL_HT_TILE;
        bl_put_tile(ht + wave * BL_HT_TILE, m, h, hb);
        __syncthreads();
        bl_second<16, false>(ring, nx, ht, m, h, acc2);
    }
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 64 * wave + 32 * c2 + 8 * g + 4 * h;
            const f32x4 bb = *reinterpret_cast<const f32x4*>(pb2 + c);
            bf16x4 yo;
            const bf16x4 x1r = *reinterpret_cast<const bf16x4*>(a.x1 + (size_t)row * BF_D + c);   // (this lane's own store, or the input)
#pragma unroll
            for (int i = 0; i < 4; ++i) yo[i] = (bf16_t)(acc2[c2][4 * g + i] + bb[i] + (float)x1r[i]);
            if (rvalid) *reinterpret_cast<bf16x4*>(a.out + (size_t)row * BF_D + c) = yo;
        }
    if (BF_TOUCH && bf_touch_fold(touched) == 0x7fc07fc1u && a.M < 0) a.mean[0] = 0.f;      // (keeps the prefetch loads alive; never true)
}

struct LeanBwd {
    int M;
    const bf16_t* x1; const bf16_t* dy;
    const float* gamma; const float* mean; const float* rstd;
    const bf16_t* z;
    const bf16x8* Wst; int wave_frags; int F;
    bf16_t* dx1;
    bf16_t* dout; int lddo;
};

// Stream per wave w: W2^T of hidden tile w; per round r: [W2^T of hidden tile 4 (r+1) + w], W1^T of its two feature tiles over the
// round's 128 hidden units in order [k'][c2]; then (CR = DI / 256 > 0) Wo^T of its DI / 4 output features in order [ks][f].
template <int ACT, int CR>
__global__ __launch_bounds__(256, 2) void block_tail_lean_bwd_kernel(LeanBwd a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * BF_ROWS;
    const int row = min(m0 + m, a.M - 1);
    const bool rvalid = m0 + m < a.M;

    // this wave's 64 features of dy (-> the shared dy tile) and of x1, the row statistics, gamma (-> LDS), the first z tile, the ring
    {
        bf16x4 dr0[2][4];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                dr0[c2][g] = *reinterpret_cast<const bf16x4*>(a.dy + (size_t)row * BF_D + 64 * wave + 32 * c2 + 8 * g + 4 * h);
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<bf16x4*>(smem + BL_YT + bf_tile_off(m, 64 * wave + 32 * c2 + 8 * g + 4 * h)) = dr0[c2][g];
    }
    if (wave == 0) reinterpret_cast<f32x4*>(smem + BL_PAR)[lane] = reinterpret_cast<const f32x4*>(a.gamma)[lane];
    const float* pgam = reinterpret_cast<const float*>(smem + BL_PAR);
    const bf16x8* zp = reinterpret_cast<const bf16x8*>(a.z) + ((size_t)(blockIdx.x * (a.F / 32) + wave) * 64 + lane) * 2;
    bf16x8 zs[2] = {zp[0], zp[1]};
    const bf16x8* nx = a.Wst + (size_t)wave * a.wave_frags * 64 + lane;
    bf16x8 ring[BF_RING];
#pragma unroll
    for (int i = 0; i < BF_RING; ++i) ring[i] = nx[i * 64];
    nx += BF_RING * 64;
    BfTouch touched;
    if (BF_TOUCH) touched = bf_touch_stream(a.Wst, 4 * a.wave_frags);
    __syncthreads();

    const int nr = a.F / 128;
    f32x16 accd[2] = {zero16(), zero16()};
    f32x16 accg = zero16();
#pragma unroll
    for (int ks = 0; ks < BF_KS; ++ks) {
        accg = mfma32(ring[ks], bl_frag256(smem + BL_YT, m, h, ks), accg);
        ring[ks] = nx[ks * 64];
    }
    nx += BF_KS * 64;
    for (int r = 0; r + 1 < nr; ++r) {
        bf16x8 hb[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) hb[i >> 3][i & 7] = (bf16_t)(accg[i] * (float)zs[i >> 3][i & 7]);
        zs[0] = zp[(size_t)(r + 1) * 4 * 128];
        zs[1] = zp[(size_t)(r + 1) * 4 * 128 + 1];
        accg = zero16();
#pragma unroll
        for (int ks = 0; ks < BF_KS; ++ks) {
            accg = mfma32(ring[16 + ks], bl_frag256(smem + BL_YT, m, h, ks), accg);
            ring[16 + ks] = nx[ks * 64];
        }
        char* ht = smem + BL_HT + (r & 1) * 4 * BL_HT_TILE;
        bl_put_tile(ht + wave * BL_HT_TILE, m, h, hb);
        __syncthreads();
        bl_second<0, true>(ring, nx + 16 * 64, ht, m, h, accd);
        nx += BF_RING * 64;
    }
    {                                                  // last round: its W1^T group sits in slots 16..31
        bf16x8 hb[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) hb[i >> 3][i & 7] = (bf16_t)(accg[i] * (float)zs[i >> 3][i & 7]);
        char* ht = smem + BL_HT + ((nr - 1) & 1) * 4 * BL_HT_TILE;
        bl_put_tile(ht + wave * BL_HT_TILE, m, h, hb);
        __syncthreads();
        bl_second<16, (CR > 0)>(ring, nx, ht, m, h, accd);
        nx += 16 * 64;
    }
    // ---- LayerNorm backward + residual branch for this wave's 64 features (operands requested now: nothing of the loop is in flight
    // any more except the next link's fragments, which are needed right afterwards anyway)
    bf16x4 xr[2][4], dr[2][4];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 64 * wave + 32 * c2 + 8 * g + 4 * h;
            dr[c2][g] = *reinterpret_cast<const bf16x4*>(a.dy + (size_t)row * BF_D + c);
            xr[c2][g] = *reinterpret_cast<const bf16x4*>(a.x1 + (size_t)row * BF_D + c);
        }
    const float mean = a.mean[row], rstd = a.rstd[row];
    float gv[2][16];                                   // (xhat is re-derived from x1 in the second pass: 32 registers less)
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 gg = *reinterpret_cast<const f32x4*>(pgam + 64 * wave + 32 * c2 + 8 * g + 4 * h);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 4 * g + i;
                gv[c2][e] = gg[i] * accd[c2][e];
                s1 += gv[c2][e];
                s2 += gv[c2][e] * (((float)xr[c2][g][i] - mean) * rstd);
            }
        }
    const float m1 = bl_rowsum(smem, 0, wave, lane, s1) * (1.f / BF_D);    // (its barrier also ends every wave's reads of the dy tile)
    const float m2 = bl_rowsum(smem, 1, wave, lane, s2) * (1.f / BF_D);
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 64 * wave + 32 * c2 + 8 * g + 4 * h;
            bf16x4 dx;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 4 * g + i;
                dx[i] = (bf16_t)((float)dr[c2][g][i] + rstd * (gv[c2][e] - m1 - (((float)xr[c2][g][i] - mean) * rstd) * m2));
            }
            if (rvalid) *reinterpret_cast<bf16x4*>(a.dx1 + (size_t)row * BF_D + c) = dx;
            if (CR > 0) *reinterpret_cast<bf16x4*>(smem + BL_YT + bf_tile_off(m, c)) = dx;
        }
    if (BF_TOUCH && bf_touch_fold(touched) == 0x7fc07fc1u && a.M < 0) a.dx1[0] = (bf16_t)0.f;  // (keeps the prefetch loads alive; never true)
    if (CR == 0) return;
    __syncthreads();
    // ---- do = dx1 Wo for this wave's DI / 4 output features (2 CR tiles of 32), stream order [ks][f]: ring positions continue
    // at slot 0 (the last W1^T group went through 16..31 and re-requested them for the second half of this link's first round)
    {
        constexpr int NF = CR > 0 ? 2 * CR : 1;        // feature tiles per wave
        f32x16 acc[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[f] = zero16();
        constexpr int TOT = 16 * NF;                   // fragments of this link
#pragma unroll
        for (int p = 0; p < TOT; ++p) {
            const int ks = p / NF, f = p % NF;
            acc[f] = mfma32(ring[p % BF_RING], bl_frag256(smem + BL_YT, m, h, ks), acc[f]);
            if (p + BF_RING < TOT) ring[p % BF_RING] = nx[(size_t)p * 64];
        }
        if (rvalid) {
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bf16x4 d = {(bf16_t)acc[f][4 * g], (bf16_t)acc[f][4 * g + 1], (bf16_t)acc[f][4 * g + 2], (bf16_t)acc[f][4 * g + 3]};
                    *reinterpret_cast<bf16x4*>(a.dout + (size_t)row * a.lddo + 32 * (wave * NF + f) + 8 * g + 4 * h) = d;
                }
        }
    }
}

template <typename K>
static int bl_prepare(K kernel) {
    if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BL_TOTAL) != hipSuccess) {
        cvft_set_error("block_lean: cannot reserve %d bytes of LDS", BL_TOTAL);
        return -2;
    }
    return 0;
}
template <int ACT, int AR>
static int launch_lean_fwd(const LeanFwd& a, hipStream_t st) {
    static int ready = 0;
    if (!ready) { if (bl_prepare(block_tail_lean_fwd_kernel<ACT, AR>)) return -2; ready = 1; }
    hipLaunchKernelGGL((block_tail_lean_fwd_kernel<ACT, AR>), dim3((a.M + BF_ROWS - 1) / BF_ROWS), dim3(256), BL_TOTAL, st, a);
    return 0;
}
template <int ACT, int CR>
static int launch_lean_bwd(const LeanBwd& a, hipStream_t st) {
    static int ready = 0;
    if (!ready) { if (bl_prepare(block_tail_lean_bwd_kernel<ACT, CR>)) return -2; ready = 1; }
    hipLaunchKernelGGL((block_tail_lean_bwd_kernel<ACT, CR>), dim3((a.M + BF_ROWS - 1) / BF_ROWS), dim3(256), BL_TOTAL, st, a);
    return 0;
}

// called by cvft_block_tail_fwd / _bwd (block_fused.hip) when args.lean != 0; arguments are already checked there
int block_tail_lean_fwd_launch(const cvft_block_tail_args* p, int DI, void* stream) {
    LeanFwd a;
    a.M = p->M; a.o = (const bf16_t*)p->o; a.ldo = p->ldo; a.x0 = (const bf16_t*)p->x0;
    a.Wst = (const bf16x8*)p->W_fwd; a.wave_frags = DI / 8 + p->F / 4; a.bo = p->bo; a.x1 = (bf16_t*)p->x1;
    a.gamma = p->gamma; a.beta = p->beta; a.eps = p->eps; a.b1 = p->b1; a.F = p->F; a.b2 = p->b2;
    a.z = (bf16_t*)p->z; a.mean = p->mean; a.rstd = p->rstd; a.out = (bf16_t*)p->out;
    const bool erf = p->act == CVFT_ACT_GELU_ERF;
    hipStream_t st = (hipStream_t)stream;
    if (DI == 0) return erf ? launch_lean_fwd<CVFT_ACT_GELU_ERF, 0>(a, st) : launch_lean_fwd<CVFT_ACT_GELU_TANH, 0>(a, st);
    if (DI == 256) return erf ? launch_lean_fwd<CVFT_ACT_GELU_ERF, 1>(a, st) : launch_lean_fwd<CVFT_ACT_GELU_TANH, 1>(a, st);
    return erf ? launch_lean_fwd<CVFT_ACT_GELU_ERF, 2>(a, st) : launch_lean_fwd<CVFT_ACT_GELU_TANH, 2>(a, st);
}
int block_tail_lean_bwd_launch(const cvft_block_tail_bwd_args* p, int DI, void* stream) {
    LeanBwd a;
    a.M = p->M; a.x1 = (const bf16_t*)p->x1; a.dy = (const bf16_t*)p->dy; a.gamma = p->gamma; a.mean = p->mean; a.rstd = p->rstd;
    a.z = (const bf16_t*)p->z; a.Wst = (const bf16x8*)p->W_bwd; a.wave_frags = p->F / 4 + p->DI / 8; a.F = p->F;
    a.dx1 = (bf16_t*)p->dx1; a.dout = (bf16_t*)p->dout; a.lddo = p->lddo;
    const bool erf = p->act == CVFT_ACT_GELU_ERF;
    hipStream_t st = (hipStream_t)stream;
    if (DI == 0) return erf ? launch_lean_bwd<CVFT_ACT_GELU_ERF, 0>(a, st) : launch_lean_bwd<CVFT_ACT_GELU_TANH, 0>(a, st);
    if (DI == 256) return erf ? launch_lean_bwd<CVFT_ACT_GELU_ERF, 1>(a, st) : launch_lean_bwd<CVFT_ACT_GELU_TANH, 1>(a, st);
    return erf ? launch_lean_bwd<CVFT_ACT_GELU_ERF, 2>(a, st) : launch_lean_bwd<CVFT_ACT_GELU_TANH, 2>(a, st);
}
