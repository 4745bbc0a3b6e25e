// ce.hip -- token-mean (label-smoothed) cross entropy with ignore index + argmax accuracy.
// Never builds the dense (n, V) true_dist / KLDiv temporaries of the reference (label_smoothing_loss.py:68-96;
// common.py:78-97): with t_c = 1-eps at the target and eps/(V-1) elsewhere the row's KL sum is
//   (1-eps) (log(1-eps) - logp_tg) + eps/(V-1) ((V-1) log(eps/(V-1)) - (sum_c logp_c - logp_tg)),   sum_c logp_c = sum_c x_c - V lse
// and its gradient softmax - t (sum_c t_c = 1).  eps = 0 is the plain NLL path, bit-for-bit the kernel it was.
#include "common.h"

// Forward: one WAVE per row, rows strided over a grid of at most 512 blocks; the three sums (nll, valid rows, correct rows) are
// carried per wave, folded per block and leave as three atomics per BLOCK.  (One block per row with three atomics per row was 72 us
// at 2 664 x 4 097: ~8 000 float atomics on three addresses serialise in one L2 channel; the row itself is 8 KB.)
// VP: 16-byte loads over the row's whole chunks (pitch and base 16-byte aligned), the V % VEC tail columns element-wise by the
// lanes 0..; two passes over the row (max / argmax, then the exponential sum), the second one from L1 / L2.
template <typename T, bool VP>
__global__ void __launch_bounds__(256) ce_fwd_kernel(int n, int V, const T* __restrict__ logits, int ld,
                                                      const int* __restrict__ target, float* __restrict__ out3,
                                                      float* __restrict__ row_lse, float eps) {
    constexpr int VEC = VP ? 16 / (int)sizeof(T) : 1;
    __shared__ float sm[4][3];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nfull = V / VEC, tail0 = nfull * VEC;          // whole chunks; first tail column (lane l takes column tail0 + l)
    float a_nll = 0.f, a_cnt = 0.f, a_ok = 0.f;
    for (int row = blockIdx.x * 4 + w; row < n; row += gridDim.x * 4) {
        const T* x = logits + (size_t)row * ld;
        float mx = -__builtin_inff(), s = 0.f, sx = 0.f;
        int am = 0x7fffffff;
#pragma unroll 4
        for (int ch = lane; ch < nfull; ch += 64) {
            T e[VEC];
            if (VP) *reinterpret_cast<uint4*>(e) = *reinterpret_cast<const uint4*>(x + ch * VEC);
            else e[0] = x[ch];
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                const float v = to_f32(e[q]);
                if (v > mx) { mx = v; am = ch * VEC + q; }
            }
        }
        float vt = -__builtin_inff();
        if (VP && tail0 + lane < V) {
            vt = to_f32(x[tail0 + lane]);
            if (vt > mx) { mx = vt; am = tail0 + lane; }
        }
        const float gmx = wave_max(mx);
        // first index attaining the max (torch.argmax tie rule): a lane's candidates were visited in increasing order
        int cand = (mx == gmx) ? am : 0x7fffffff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
#pragma unroll 4
        for (int ch = lane; ch < nfull; ch += 64) {
            T e[VEC];
            if (VP) *reinterpret_cast<uint4*>(e) = *reinterpret_cast<const uint4*>(x + ch * VEC);
            else e[0] = x[ch];
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                const float v = to_f32(e[q]);
                s += __expf(v - gmx);
                sx += v;
            }
        }
        if (VP && tail0 + lane < V) {
            s += __expf(vt - gmx);
            sx += vt;
        }
        s = wave_sum(s);
        if (eps > 0.f) sx = wave_sum(sx);
        const float lse = gmx + logf(s);
        const int tg = target[row];
        if (lane == 0) row_lse[row] = lse;
        if (tg >= 0) {            // wave-uniform
            const float xt = to_f32(x[tg]);
            float nll = lse - xt;
            if (eps > 0.f) {
                const float conf = 1.f - eps, sm_ = eps / (float)(V - 1), lp_tg = xt - lse;
                const float lp_rest = (sx - (float)V * lse) - lp_tg;                 // sum over c != tg of logp_c
                nll = (conf > 0.f ? conf * (logf(conf) - lp_tg) : 0.f) + sm_ * ((float)(V - 1) * logf(sm_) - lp_rest);
            }
            a_nll += nll;
            a_cnt += 1.f;
            a_ok += (cand == tg) ? 1.f : 0.f;
        }
    }
    if (lane == 0) { sm[w][0] = a_nll; sm[w][1] = a_cnt; sm[w][2] = a_ok; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const float v = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
        if (v != 0.f) atomicAdd(&out3[threadIdx.x], v);
    }
}

// Backward: one wave per row, 16-byte loads and stores on the VP path (a chunk that straddles V gets zeros in its pad columns, which
// is what the pitch padding holds anyway).
template <typename T, bool VP>
__global__ void __launch_bounds__(256) ce_bwd_kernel(int n, int V, const T* __restrict__ logits, int ld,
                                                      const int* __restrict__ target, const float* __restrict__ row_lse,
                                                      const float* __restrict__ gscale, T* __restrict__ dl, int ldd, float eps) {
    constexpr int VEC = VP ? 16 / (int)sizeof(T) : 1;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const T* x = logits + (size_t)row * ld;
    T* d = dl + (size_t)row * ldd;
    const int tg = target[row];
    const int nch = (V + VEC - 1) / VEC;
    const float g = tg < 0 ? 0.f : gscale[0], lse = row_lse[row];
    const float t_on = 1.f - eps, t_off = eps > 0.f ? eps / (float)(V - 1) : 0.f;
    for (int ch = lane; ch < nch; ch += 64) {
        T e[VEC], o[VEC];
        if (VP) *reinterpret_cast<uint4*>(e) = *reinterpret_cast<const uint4*>(x + ch * VEC);
        else e[0] = x[ch];
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            const int c = ch * VEC + q;
            const float pr = __expf(to_f32(e[q]) - lse);
            o[q] = from_f32<T>((tg >= 0 && c < V) ? g * (pr - (c == tg ? t_on : t_off)) : 0.f);
        }
        if (VP) *reinterpret_cast<uint4*>(d + ch * VEC) = *reinterpret_cast<const uint4*>(o);
        else d[ch] = o[0];
    }
}

extern "C" int cvft_ce_fwd(int dtype, int n, int V, const void* logits, int ld, const int32_t* target, float* out3,
                           float* row_lse, float smoothing, void* stream) {
    CVFT_CHECK_ARG(dtype == CVFT_F32 || dtype == CVFT_BF16, "cvft_ce_fwd: bad dtype");
    CVFT_CHECK_ARG(n >= 0 && V > 1 && ld >= V && logits && target && out3 && row_lse, "cvft_ce_fwd: bad args");
    CVFT_CHECK_ARG(smoothing >= 0.f && smoothing <= 1.f, "cvft_ce_fwd: smoothing outside [0, 1]");
    if (n == 0) return 0;
    const int esz = dtype == CVFT_F32 ? 4 : 2, vec = 16 / esz;
    const bool vp = (reinterpret_cast<uintptr_t>(logits) & 15) == 0 && ld % vec == 0;          // 16-byte loads: aligned base and pitch
    // grid cap 256 / 512 / 1024 / 2048 blocks: 33.5 / 21.2 / 23.1 / 26.7 us at 5 328 x 4 097 bf16 (tools/bench_ce.py; more blocks = more atomics)
    const dim3 grid((unsigned)((n + 3) / 4 < 512 ? (n + 3) / 4 : 512));
#define CE_FWD(TT, VPv) hipLaunchKernelGGL((ce_fwd_kernel<TT, VPv>), grid, dim3(256), 0, (hipStream_t)stream, n, V, (const TT*)logits, ld, target, \
                                           out3, row_lse, smoothing)
    if (dtype == CVFT_F32) { if (vp) CE_FWD(float, true); else CE_FWD(float, false); }
    else { if (vp) CE_FWD(bf16_t, true); else CE_FWD(bf16_t, false); }
#undef CE_FWD
    CVFT_LAUNCH_CHECK("cvft_ce_fwd");
    return 0;
}
extern "C" int cvft_ce_bwd(int dtype, int n, int V, const void* logits, int ld, const int32_t* target,
                           const float* row_lse, const float* gscale, void* dlogits, int ldd, float smoothing, void* stream) {
    CVFT_CHECK_ARG(dtype == CVFT_F32 || dtype == CVFT_BF16, "cvft_ce_bwd: bad dtype");
    CVFT_CHECK_ARG(smoothing >= 0.f && smoothing <= 1.f, "cvft_ce_bwd: smoothing outside [0, 1]");
    CVFT_CHECK_ARG(n >= 0 && V > 1 && ld >= V && ldd >= V && logits && target && row_lse && gscale && dlogits, "cvft_ce_bwd: bad args");
    if (n == 0) return 0;
    const int vec = dtype == CVFT_F32 ? 4 : 8, vr = (V + vec - 1) / vec * vec;
    const bool vp = ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(dlogits)) & 15) == 0 && ld % vec == 0 && ldd % vec == 0 &&
                    ld >= vr && ldd >= vr;
    const dim3 grid((unsigned)((n + 3) / 4));
#define CE_BWD(TT, VPv) hipLaunchKernelGGL((ce_bwd_kernel<TT, VPv>), grid, dim3(256), 0, (hipStream_t)stream, n, V, (const TT*)logits, ld, target, \
                                           row_lse, gscale, (TT*)dlogits, ldd, smoothing)
    if (dtype == CVFT_F32) { if (vp) CE_BWD(float, true); else CE_BWD(float, false); }
    else { if (vp) CE_BWD(bf16_t, true); else CE_BWD(bf16_t, false); }
#undef CE_BWD
    CVFT_LAUNCH_CHECK("cvft_ce_bwd");
    return 0;
}
