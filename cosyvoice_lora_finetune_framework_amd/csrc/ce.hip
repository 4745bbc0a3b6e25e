// ce.hip -- token-mean (label-smoothed) cross entropy with ignore index + argmax accuracy, one block per row.
// Never builds the dense (n, V) true_dist / KLDiv temporaries of the reference (label_smoothing_loss.py:68-96;
// common.py:78-97): with t_c = 1-eps at the target and eps/(V-1) elsewhere the row's KL sum is
//   (1-eps) (log(1-eps) - logp_tg) + eps/(V-1) ((V-1) log(eps/(V-1)) - (sum_c logp_c - logp_tg)),   sum_c logp_c = sum_c x_c - V lse
// and its gradient softmax - t (sum_c t_c = 1).  eps = 0 is the plain NLL path, bit-for-bit the kernel it was.
#include "common.h"

__device__ __forceinline__ float block_max(float v, float* sm) {
    v = wave_max(v);
    int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = blockDim.x >> 6;
    __syncthreads();
    if (l == 0) sm[w] = v;
    __syncthreads();
    float r = sm[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, sm[i]);
    return r;
}

template <typename T>
__global__ void __launch_bounds__(256) ce_fwd_kernel(int n, int V, const T* __restrict__ logits, int ld,
                                                      const int* __restrict__ target, float* __restrict__ out3,
                                                      float* __restrict__ row_lse, float eps) {
    __shared__ float sm[16];
    __shared__ int smi[4];
    const int row = blockIdx.x;
    const T* x = logits + (size_t)row * ld;
    float mx = -__builtin_inff();
    int am = 0x7fffffff;
    for (int c = threadIdx.x; c < V; c += 256) {
        float v = to_f32(x[c]);
        if (v > mx) { mx = v; am = c; }
    }
    const float gmx = block_max(mx, sm);
    // first index attaining the max (torch.argmax tie rule)
    int cand = (mx == gmx) ? am : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) smi[threadIdx.x >> 6] = cand;
    __syncthreads();
    const int amax = min(min(smi[0], smi[1]), min(smi[2], smi[3]));
    float s = 0.f, sx = 0.f;
    for (int c = threadIdx.x; c < V; c += 256) {
        const float v = to_f32(x[c]);
        s += __expf(v - gmx);
        sx += v;
    }
    s = block_sum(s, sm);
    if (eps > 0.f) {          // block-uniform
        __syncthreads();
        sx = block_sum(sx, sm);
    }
    if (threadIdx.x == 0) {
        const float lse = gmx + logf(s);
        row_lse[row] = lse;
        const int tg = target[row];
        if (tg >= 0) {
            float nll = lse - to_f32(x[tg]);
            if (eps > 0.f) {
                const float conf = 1.f - eps, sm_ = eps / (float)(V - 1), lp_tg = to_f32(x[tg]) - lse;
                const float lp_rest = (sx - (float)V * lse) - lp_tg;                 // sum over c != tg of logp_c
                nll = (conf > 0.f ? conf * (logf(conf) - lp_tg) : 0.f) + sm_ * ((float)(V - 1) * logf(sm_) - lp_rest);
            }
            atomicAdd(&out3[0], nll);
            atomicAdd(&out3[1], 1.f);
            if (amax == tg) atomicAdd(&out3[2], 1.f);
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) ce_bwd_kernel(int n, int V, const T* __restrict__ logits, int ld,
                                                      const int* __restrict__ target, const float* __restrict__ row_lse,
                                                      const float* __restrict__ gscale, T* __restrict__ dl, int ldd, float eps) {
    const int row = blockIdx.x;
    const T* x = logits + (size_t)row * ld;
    T* d = dl + (size_t)row * ldd;
    const int tg = target[row];
    if (tg < 0) {
        for (int c = threadIdx.x; c < V; c += 256) d[c] = from_f32<T>(0.f);
        return;
    }
    const float g = gscale[0], lse = row_lse[row];
    const float t_on = 1.f - eps, t_off = eps > 0.f ? eps / (float)(V - 1) : 0.f;
    for (int c = threadIdx.x; c < V; c += 256) {
        float pr = __expf(to_f32(x[c]) - lse);
        d[c] = from_f32<T>(g * (pr - (c == tg ? t_on : t_off)));
    }
}

extern "C" int cvft_ce_fwd(int dtype, int n, int V, const void* logits, int ld, const int32_t* target, float* out3,
                           float* row_lse, float smoothing, void* stream) {
    CVFT_CHECK_ARG(dtype == CVFT_F32 || dtype == CVFT_BF16, "cvft_ce_fwd: bad dtype");
    CVFT_CHECK_ARG(n >= 0 && V > 1 && ld >= V && logits && target && out3 && row_lse, "cvft_ce_fwd: bad args");
    CVFT_CHECK_ARG(smoothing >= 0.f && smoothing <= 1.f, "cvft_ce_fwd: smoothing outside [0, 1]");
    if (n == 0) return 0;
    if (dtype == CVFT_F32)
        hipLaunchKernelGGL((ce_fwd_kernel<float>), dim3(n), dim3(256), 0, (hipStream_t)stream, n, V, (const float*)logits, ld, target, out3, row_lse, smoothing);
    else
        hipLaunchKernelGGL((ce_fwd_kernel<bf16_t>), dim3(n), dim3(256), 0, (hipStream_t)stream, n, V, (const bf16_t*)logits, ld, target, out3, row_lse, smoothing);
    CVFT_LAUNCH_CHECK("cvft_ce_fwd");
    return 0;
}
extern "C" int cvft_ce_bwd(int dtype, int n, int V, const void* logits, int ld, const int32_t* target,
                           const float* row_lse, const float* gscale, void* dlogits, int ldd, float smoothing, void* stream) {
    CVFT_CHECK_ARG(dtype == CVFT_F32 || dtype == CVFT_BF16, "cvft_ce_bwd: bad dtype");
    CVFT_CHECK_ARG(smoothing >= 0.f && smoothing <= 1.f, "cvft_ce_bwd: smoothing outside [0, 1]");
    CVFT_CHECK_ARG(n >= 0 && V > 1 && ld >= V && ldd >= V && logits && target && row_lse && gscale && dlogits, "cvft_ce_bwd: bad args");
    if (n == 0) return 0;
    if (dtype == CVFT_F32)
        hipLaunchKernelGGL((ce_bwd_kernel<float>), dim3(n), dim3(256), 0, (hipStream_t)stream, n, V, (const float*)logits, ld, target, row_lse, gscale, (float*)dlogits, ldd, smoothing);
    else
        hipLaunchKernelGGL((ce_bwd_kernel<bf16_t>), dim3(n), dim3(256), 0, (hipStream_t)stream, n, V, (const bf16_t*)logits, ld, target, row_lse, gscale, (bf16_t*)dlogits, ldd, smoothing);
    CVFT_LAUNCH_CHECK("cvft_ce_bwd");
    return 0;
}
