// elementwise.hip -- small HBM-bound fused ops of the hot path (gathers, CFM prepare,
// masked MSE, interpolation, depthwise conv, time embedding, optimiser pieces).
// All are grid-stride, coalesced along the channel (fastest) axis.
#include <type_traits>
#include "common.h"

static inline unsigned ew_grid(size_t total) {
    size_t g = (total + 255) / 256;
    return (unsigned)(g > 8192 ? 8192 : (g == 0 ? 1 : g));
}
#define EW_LOOP(i, total) for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (total); i += (size_t)gridDim.x * 256)
#define CHECK_DTYPE(name, dtype) CVFT_CHECK_ARG((dtype) == CVFT_F32 || (dtype) == CVFT_BF16, name ": bad dtype")

// ---------------------------------------------------------------- embedding gather
template <typename T>
__global__ void embed_gather_kernel(int B, int L, int D, const int64_t* __restrict__ tok, const int* __restrict__ len,
                                    const T* __restrict__ table, T* __restrict__ out) {
    const size_t total = (size_t)B * L * D;
    EW_LOOP(i, total) {
        int d = (int)(i % D);
        size_t bl = i / D;
        int l = (int)(bl % L), b = (int)(bl / L);
        int64_t t = tok[bl];
        if (t < 0) t = 0;
        T v = table[(size_t)t * D + d];
        if (len && l >= len[b]) v = from_f32<T>(0.f);
        out[i] = v;
    }
}
extern "C" int cvft_embed_gather(int dtype, int B, int L, int D, const int64_t* tok, const int32_t* len,
                                 const void* table, void* out, void* stream) {
    CHECK_DTYPE("cvft_embed_gather", dtype);
    CVFT_CHECK_ARG(B > 0 && L > 0 && D > 0 && tok && table && out, "cvft_embed_gather: bad args");
    size_t total = (size_t)B * L * D;
    if (dtype == CVFT_F32)
        hipLaunchKernelGGL((embed_gather_kernel<float>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, B, L, D, tok, len,
                           (const float*)table, (float*)out);
    else
        hipLaunchKernelGGL((embed_gather_kernel<bf16_t>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, B, L, D, tok, len,
                           (const bf16_t*)table, (bf16_t*)out);
    CVFT_LAUNCH_CHECK("cvft_embed_gather");
    return 0;
}

// ---------------------------------------------------------------- ragged row gather / scatter
template <typename T>
__global__ void gather_rows_kernel(int n, int D, const int* __restrict__ idx, const T* __restrict__ src, float fill,
                                   T* __restrict__ out) {
    const size_t total = (size_t)n * D;
    EW_LOOP(i, total) {
        int d = (int)(i % D);
        int r = (int)(i / D);
        int s = idx[r];
        out[i] = s >= 0 ? src[(size_t)s * D + d] : from_f32<T>(fill);
    }
}
template <typename T>
__global__ void scatter_rows_kernel(int n, int D, const int* __restrict__ idx, const T* __restrict__ dout,
                                    T* __restrict__ dsrc) {
    const size_t total = (size_t)n * D;
    EW_LOOP(i, total) {
        int d = (int)(i % D);
        int r = (int)(i / D);
        int s = idx[r];
        if (s >= 0) dsrc[(size_t)s * D + d] = dout[i];
    }
}
extern "C" int cvft_gather_rows(int dtype, int n, int D, const int32_t* idx, const void* src, float fill, void* out,
                                void* stream) {
    CHECK_DTYPE("cvft_gather_rows", dtype);
    CVFT_CHECK_ARG(n >= 0 && D > 0 && idx && src && out, "cvft_gather_rows: bad args");
    if (n == 0) return 0;
    size_t total = (size_t)n * D;
    if (dtype == CVFT_F32)
        hipLaunchKernelGGL((gather_rows_kernel<float>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, n, D, idx,
                           (const float*)src, fill, (float*)out);
    else
        hipLaunchKernelGGL((gather_rows_kernel<bf16_t>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, n, D, idx,
                           (const bf16_t*)src, fill, (bf16_t*)out);
    CVFT_LAUNCH_CHECK("cvft_gather_rows");
    return 0;
}
extern "C" int cvft_scatter_rows(int dtype, int n, int D, const int32_t* idx, const void* dout, void* dsrc, void* stream) {
    CHECK_DTYPE("cvft_scatter_rows", dtype);
    CVFT_CHECK_ARG(n >= 0 && D > 0 && idx && dout && dsrc, "cvft_scatter_rows: bad args");
    if (n == 0) return 0;
    size_t total = (size_t)n * D;
    if (dtype == CVFT_F32)
        hipLaunchKernelGGL((scatter_rows_kernel<float>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, n, D, idx,
                           (const float*)dout, (float*)dsrc);
    else
        hipLaunchKernelGGL((scatter_rows_kernel<bf16_t>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, n, D, idx,
                           (const bf16_t*)dout, (bf16_t*)dsrc);
    CVFT_LAUNCH_CHECK("cvft_scatter_rows");
    return 0;
}

// ---------------------------------------------------------------- row L2 normalise (one wave per row)
template <typename T>
__global__ void l2norm_rows_kernel(int rows, int D, const float* __restrict__ x, T* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * D;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += xr[c] * xr[c];
    const float nrm = fmaxf(sqrtf(wave_sum(s)), 1e-12f);
    for (int c = lane; c < D; c += 64) y[(size_t)row * D + c] = from_f32<T>(xr[c] / nrm);
}
extern "C" int cvft_l2norm_rows(int dtype, int rows, int D, const float* x, void* y, void* stream) {
    CHECK_DTYPE("cvft_l2norm_rows", dtype);
    CVFT_CHECK_ARG(rows > 0 && D > 0 && x && y, "cvft_l2norm_rows: bad args");
    if (dtype == CVFT_F32)
        hipLaunchKernelGGL((l2norm_rows_kernel<float>), dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, rows, D, x, (float*)y);
    else
        hipLaunchKernelGGL((l2norm_rows_kernel<bf16_t>), dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, rows, D, x, (bf16_t*)y);
    CVFT_LAUNCH_CHECK("cvft_l2norm_rows");
    return 0;
}

// ---------------------------------------------------------------- sinusoidal time embedding (scale 1000)
template <typename T>
__global__ void time_embed_kernel(int B, int dim, const float* __restrict__ t, const float* __restrict__ freqs,
                                  float scale, T* __restrict__ out) {
    const int half = dim / 2;
    const size_t total = (size_t)B * dim;
    EW_LOOP(i, total) {
        int d = (int)(i % dim), b = (int)(i / dim);
        int k = d < half ? d : d - half;
        float e = scale * t[b] * freqs[k];
        out[i] = from_f32<T>(d < half ? sinf(e) : cosf(e));
    }
}
extern "C" int cvft_time_embed(int dtype, int B, int dim, const float* t, const float* freqs, float scale, void* out,
                               void* stream) {
    CHECK_DTYPE("cvft_time_embed", dtype);
    CVFT_CHECK_ARG(B > 0 && dim >= 4 && dim % 2 == 0 && t && freqs && out, "cvft_time_embed: bad args");
    if (dtype == CVFT_F32)
        hipLaunchKernelGGL((time_embed_kernel<float>), dim3(ew_grid((size_t)B * dim)), dim3(256), 0, (hipStream_t)stream, B, dim, t, freqs, scale, (float*)out);
    else
        hipLaunchKernelGGL((time_embed_kernel<bf16_t>), dim3(ew_grid((size_t)B * dim)), dim3(256), 0, (hipStream_t)stream, B, dim, t, freqs, scale, (bf16_t*)out);
    CVFT_LAUNCH_CHECK("cvft_time_embed");
    return 0;
}

// ---------------------------------------------------------------- elementwise activation
template <typename T>
__global__ void act_fwd_kernel(size_t n, int act, const T* __restrict__ x, T* __restrict__ y) {
    EW_LOOP(i, n) y[i] = from_f32<T>(act_apply(act, to_f32(x[i])));
}
extern "C" int cvft_act_fwd(int dtype, int64_t n, int act, const void* x, void* y, void* stream) {
    CHECK_DTYPE("cvft_act_fwd", dtype);
    CVFT_CHECK_ARG(n >= 0 && x && y, "cvft_act_fwd: bad args");
    if (n == 0) return 0;
    if (dtype == CVFT_F32)
        hipLaunchKernelGGL((act_fwd_kernel<float>), dim3(ew_grid((size_t)n)), dim3(256), 0, (hipStream_t)stream, (size_t)n, act, (const float*)x, (float*)y);
    else
        hipLaunchKernelGGL((act_fwd_kernel<bf16_t>), dim3(ew_grid((size_t)n)), dim3(256), 0, (hipStream_t)stream, (size_t)n, act, (const bf16_t*)x, (bf16_t*)y);
    CVFT_LAUNCH_CHECK("cvft_act_fwd");
    return 0;
}

template <typename T>
__global__ void act_bwd_kernel(size_t n, int act, const T* __restrict__ z, const T* __restrict__ dy, T* __restrict__ dz) {
    EW_LOOP(i, n) dz[i] = from_f32<T>(to_f32(dy[i]) * act_grad(act, to_f32(z[i])));
}
extern "C" int cvft_act_bwd(int dtype, int64_t n, int act, const void* z, const void* dy, void* dz, void* stream) {
    CHECK_DTYPE("cvft_act_bwd", dtype);
    CVFT_CHECK_ARG(n >= 0 && z && dy && dz, "cvft_act_bwd: bad args");
    if (n == 0) return 0;
    if (dtype == CVFT_F32)
        hipLaunchKernelGGL((act_bwd_kernel<float>), dim3(ew_grid((size_t)n)), dim3(256), 0, (hipStream_t)stream, (size_t)n, act, (const float*)z, (const float*)dy, (float*)dz);
    else
        hipLaunchKernelGGL((act_bwd_kernel<bf16_t>), dim3(ew_grid((size_t)n)), dim3(256), 0, (hipStream_t)stream, (size_t)n, act, (const bf16_t*)z, (const bf16_t*)dy, (bf16_t*)dz);
    CVFT_LAUNCH_CHECK("cvft_act_bwd");
    return 0;
}

// ---------------------------------------------------------------- CFM prepare
template <typename T>
__global__ void cfm_prepare_kernel(int B, int T_, const float* __restrict__ feat, const float* __restrict__ z,
                                   const float* __restrict__ t_raw, const float* __restrict__ keep,
                                   const T* __restrict__ mu, const T* __restrict__ spk, const T* __restrict__ cond,
                                   float mel_mean, float mel_std, float sigma_min, int cosine, T* __restrict__ xin,
                                   float* __restrict__ u, float* __restrict__ tout) {
    const size_t total = (size_t)B * T_ * 80;
    EW_LOOP(i, total) {
        int c = (int)(i % 80);
        size_t bt = i / 80;
        int b = (int)(bt / T_);
        const float t = cosine ? 1.f - cosf(t_raw[b] * 0.5f * 3.14159265358979323846f) : t_raw[b];     // t_scheduler (flow_matching.py:176)
        const float x1 = (feat[i] - mel_mean) / mel_std;
        const float zz = z[i];
        const float y = (1.f - (1.f - sigma_min) * t) * zz + t * x1;
        u[i] = x1 - (1.f - sigma_min) * zz;
        const float kp = keep[b];
        T* row = xin + bt * 320;
        row[c] = from_f32<T>(y);
        row[80 + c] = from_f32<T>(to_f32(mu[i]) * kp);
        row[160 + c] = from_f32<T>(to_f32(spk[(size_t)b * 80 + c]) * kp);
        row[240 + c] = from_f32<T>(cond ? to_f32(cond[i]) * kp : 0.f);
        if (c == 0 && (bt % T_) == 0) tout[b] = t;
    }
}
extern "C" int cvft_cfm_prepare(int dtype, int B, int T, const float* feat, const float* z, const float* t_raw,
                                const float* cfg_keep, const void* mu, const void* spk, const void* cond, float mel_mean,
                                float mel_std, float sigma_min, int t_cosine, void* xin, float* u, float* t, void* stream) {
    CHECK_DTYPE("cvft_cfm_prepare", dtype);
    CVFT_CHECK_ARG(B > 0 && T > 0 && feat && z && t_raw && cfg_keep && mu && spk && xin && u && t, "cvft_cfm_prepare: bad args");
    size_t total = (size_t)B * T * 80;
    if (dtype == CVFT_F32)
        hipLaunchKernelGGL((cfm_prepare_kernel<float>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, B, T, feat, z,
                           t_raw, cfg_keep, (const float*)mu, (const float*)spk, (const float*)cond, mel_mean, mel_std, sigma_min, t_cosine, (float*)xin, u, t);
    else
        hipLaunchKernelGGL((cfm_prepare_kernel<bf16_t>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, B, T, feat, z,
                           t_raw, cfg_keep, (const bf16_t*)mu, (const bf16_t*)spk, (const bf16_t*)cond, mel_mean, mel_std, sigma_min, t_cosine, (bf16_t*)xin, u, t);
    CVFT_LAUNCH_CHECK("cvft_cfm_prepare");
    return 0;
}

// ---------------------------------------------------------------- masked MSE
template <typename T>
__global__ void __launch_bounds__(256) masked_mse_fwd_kernel(int B, int T_, int C, const T* __restrict__ pred,
                                                              const float* __restrict__ u, const int* __restrict__ len,
                                                              const float* __restrict__ w, float* __restrict__ loss_sum) {
    __shared__ float sm[16];
    const size_t total = (size_t)B * T_ * C;
    float s = 0.f;
    EW_LOOP(i, total) {
        size_t bt = i / C;
        int t = (int)(bt % T_), b = (int)(bt / T_);
        if (!len || t < len[b]) {
            float d = to_f32(pred[i]) - u[i];
            if (w) d *= w[bt];              // per-frame loss weight (prompt region 0, boundary frames > 1; flow_model.py:179-202)
            s += d * d;
        }
    }
    s = block_sum(s, sm);
    if (threadIdx.x == 0) atomicAdd(loss_sum, s);
}
template <typename T>
__global__ void masked_mse_bwd_kernel(int B, int T_, int C, const T* __restrict__ pred, const float* __restrict__ u,
                                      const int* __restrict__ len, const float* __restrict__ w, const float* __restrict__ gscale,
                                      T* __restrict__ dpred) {
    const size_t total = (size_t)B * T_ * C;
    const float g = 2.f * gscale[0];
    EW_LOOP(i, total) {
        size_t bt = i / C;
        int t = (int)(bt % T_), b = (int)(bt / T_);
        float d = 0.f;
        if (!len || t < len[b]) d = g * (to_f32(pred[i]) - u[i]) * (w ? w[bt] * w[bt] : 1.f);
        dpred[i] = from_f32<T>(d);
    }
}
extern "C" int cvft_masked_mse_fwd(int dtype, int B, int T, int C, const void* pred, const float* u, const int32_t* len,
                                   const float* w, float* loss_sum, void* stream) {
    CHECK_DTYPE("cvft_masked_mse_fwd", dtype);
    CVFT_CHECK_ARG(B > 0 && T > 0 && C > 0 && pred && u && loss_sum, "cvft_masked_mse_fwd: bad args");
    size_t total = (size_t)B * T * C;
    unsigned g = ew_grid(total);
    if (g > 1024) g = 1024;
    if (dtype == CVFT_F32)
        hipLaunchKernelGGL((masked_mse_fwd_kernel<float>), dim3(g), dim3(256), 0, (hipStream_t)stream, B, T, C, (const float*)pred, u, len, w, loss_sum);
    else
        hipLaunchKernelGGL((masked_mse_fwd_kernel<bf16_t>), dim3(g), dim3(256), 0, (hipStream_t)stream, B, T, C, (const bf16_t*)pred, u, len, w, loss_sum);
    CVFT_LAUNCH_CHECK("cvft_masked_mse_fwd");
    return 0;
}
extern "C" int cvft_masked_mse_bwd(int dtype, int B, int T, int C, const void* pred, const float* u, const int32_t* len,
                                   const float* w, const float* gscale, void* dpred, void* stream) {
    CHECK_DTYPE("cvft_masked_mse_bwd", dtype);
    CVFT_CHECK_ARG(B > 0 && T > 0 && C > 0 && pred && u && gscale && dpred, "cvft_masked_mse_bwd: bad args");
    size_t total = (size_t)B * T * C;
    if (dtype == CVFT_F32)
        hipLaunchKernelGGL((masked_mse_bwd_kernel<float>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, B, T, C, (const float*)pred, u, len, w, gscale, (float*)dpred);
    else
        hipLaunchKernelGGL((masked_mse_bwd_kernel<bf16_t>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, B, T, C, (const bf16_t*)pred, u, len, w, gscale, (bf16_t*)dpred);
    CVFT_LAUNCH_CHECK("cvft_masked_mse_bwd");
    return 0;
}

// ---------------------------------------------------------------- linear interpolation along time
// torch upsample_linear1d, align_corners=False: src = max(scale*(dst+0.5)-0.5, 0), scale = Lin/Lout
__device__ __forceinline__ void interp_src(int j, float scale, int Lin, int& i0, int& i1, float& w0, float& w1) {
    float src = scale * ((float)j + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    if (i0 > Lin - 1) i0 = Lin - 1;
    i1 = i0 + ((i0 < Lin - 1) ? 1 : 0);
    w1 = src - (float)i0;
    w0 = 1.f - w1;
}
template <typename T>
__global__ void interp_fwd_kernel(int B, int Lin, int Lout, int C, const T* __restrict__ x, T* __restrict__ y,
                                  const int* __restrict__ eff) {
    // eff (cvft.h): the exact-shape batch's (Lin, Lout) when the tensors are padded to buckets -- they set the scale and
    // the clamp; the tensor dims only the strides
    const size_t total = (size_t)B * Lout * C;
    const int Li = eff ? min(eff[0], Lin) : Lin, Lo = eff ? min(eff[1], Lout) : Lout;
    const float scale = (float)Li / (float)Lo;
    EW_LOOP(i, total) {
        int c = (int)(i % C);
        size_t bj = i / C;
        int j = (int)(bj % Lout), b = (int)(bj / Lout);
        if (j >= Lo) { y[i] = from_f32<T>(0.f); continue; }
        int i0, i1;
        float w0, w1;
        interp_src(j, scale, Li, i0, i1, w0, w1);
        const T* xb = x + (size_t)b * Lin * C;
        y[i] = from_f32<T>(w0 * to_f32(xb[(size_t)i0 * C + c]) + w1 * to_f32(xb[(size_t)i1 * C + c]));
    }
}
template <typename T>
__global__ void interp_bwd_kernel(int B, int Lin, int Lout, int C, const T* __restrict__ dy, T* __restrict__ dx,
                                  const int* __restrict__ eff) {
    const size_t total = (size_t)B * Lin * C;
    const int Li = eff ? min(eff[0], Lin) : Lin, Lo = eff ? min(eff[1], Lout) : Lout;
    const float scale = (float)Li / (float)Lo;
    const float inv = (float)Lo / (float)Li;
    EW_LOOP(i, total) {
        int c = (int)(i % C);
        size_t bi = i / C;
        int ii = (int)(bi % Lin), b = (int)(bi / Lin);
        if (ii >= Li) { dx[i] = from_f32<T>(0.f); continue; }
        // candidate outputs: src in (ii-1, ii+1)  =>  j in ((ii-0.5)*inv-0.5, (ii+1.5)*inv-0.5); widen by one
        int jlo = (int)floorf(((float)ii - 0.5f) * inv - 0.5f) - 1;
        int jhi = (int)ceilf(((float)ii + 1.5f) * inv - 0.5f) + 1;
        if (jlo < 0) jlo = 0;
        if (jhi > Lo - 1) jhi = Lo - 1;
        const T* dyb = dy + (size_t)b * Lout * C;
        float s = 0.f;
        for (int j = jlo; j <= jhi; ++j) {
            int i0, i1;
            float w0, w1;
            interp_src(j, scale, Li, i0, i1, w0, w1);
            float g = to_f32(dyb[(size_t)j * C + c]);
            if (i0 == ii) s += w0 * g;
            if (i1 == ii) s += w1 * g;
        }
        dx[i] = from_f32<T>(s);
    }
}
extern "C" int cvft_interp_linear_fwd(int dtype, int B, int Lin, int Lout, int C, const void* x, void* y, const int32_t* eff, void* stream) {
    CHECK_DTYPE("cvft_interp_linear_fwd", dtype);
    CVFT_CHECK_ARG(B > 0 && Lin > 0 && Lout > 0 && C > 0 && x && y, "cvft_interp_linear_fwd: bad args");
    size_t total = (size_t)B * Lout * C;
    if (dtype == CVFT_F32)
        hipLaunchKernelGGL((interp_fwd_kernel<float>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, B, Lin, Lout, C, (const float*)x, (float*)y, eff);
    else
        hipLaunchKernelGGL((interp_fwd_kernel<bf16_t>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, B, Lin, Lout, C, (const bf16_t*)x, (bf16_t*)y, eff);
    CVFT_LAUNCH_CHECK("cvft_interp_linear_fwd");
    return 0;
}
extern "C" int cvft_interp_linear_bwd(int dtype, int B, int Lin, int Lout, int C, const void* dy, void* dx, const int32_t* eff, void* stream) {
    CHECK_DTYPE("cvft_interp_linear_bwd", dtype);
    CVFT_CHECK_ARG(B > 0 && Lin > 0 && Lout > 0 && C > 0 && dy && dx, "cvft_interp_linear_bwd: bad args");
    size_t total = (size_t)B * Lin * C;
    if (dtype == CVFT_F32)
        hipLaunchKernelGGL((interp_bwd_kernel<float>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, B, Lin, Lout, C, (const float*)dy, (float*)dx, eff);
    else
        hipLaunchKernelGGL((interp_bwd_kernel<bf16_t>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, B, Lin, Lout, C, (const bf16_t*)dy, (bf16_t*)dx, eff);
    CVFT_LAUNCH_CHECK("cvft_interp_linear_bwd");
    return 0;
}

// ---------------------------------------------------------------- depthwise conv1d (groups = C), channel-last
// One block = 64 frames x 64 channels; the (64 + Kw - 1) x 64 input window is staged in LDS so every
// frame is read from HBM once, coalesced along channels.
template <typename T, bool BWD>
__global__ void __launch_bounds__(256) dwconv_kernel(int B, int T_, int C, int Kw, int pad_left, const T* __restrict__ x,
                                                      const float* __restrict__ w, const float* __restrict__ bias,
                                                      const int* __restrict__ len, T* __restrict__ y) {
    extern __shared__ float win[];   // (64 + Kw - 1) x 64
    const int tt = (T_ + 63) / 64;
    const int b = blockIdx.x / tt, t0 = (blockIdx.x % tt) * 64, c0 = blockIdx.y * 64;
    const int lb = len ? len[b] : T_;
    const int rows = 64 + Kw - 1;
    // fwd: y[t] = sum_k w[k] * xm[t + k - pad_left];  bwd: dx[t] = m[t] * sum_k w[k] * dy[t - k + pad_left]
    const int base = BWD ? t0 - (Kw - 1 - pad_left) : t0 - pad_left;
    for (int e = threadIdx.x; e < rows * 64; e += 256) {
        int r = e >> 6, c = e & 63;
        int t = base + r;
        float v = 0.f;
        if (t >= 0 && t < T_ && c0 + c < C && (BWD || t < lb)) v = to_f32(x[((size_t)b * T_ + t) * C + c0 + c]);
        win[e] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        int r = e >> 6, c = e & 63;
        int t = t0 + r;
        if (t >= T_ || c0 + c >= C) continue;
        const float* wc = w + (size_t)(c0 + c) * Kw;
        float s = (!BWD && bias) ? bias[c0 + c] : 0.f;
        for (int k = 0; k < Kw; ++k) s += wc[k] * win[(BWD ? (r + Kw - 1 - k) : (r + k)) * 64 + c];
        if (BWD && t >= lb) s = 0.f;
        y[((size_t)b * T_ + t) * C + c0 + c] = from_f32<T>(s);
    }
}
extern "C" int cvft_dwconv1d_fwd(int dtype, int B, int T, int C, int Kw, int pad_left, const void* x, const float* w,
                                 const float* bias, const int32_t* len, void* y, void* stream) {
    CHECK_DTYPE("cvft_dwconv1d_fwd", dtype);
    CVFT_CHECK_ARG(B > 0 && T > 0 && C > 0 && Kw > 0 && Kw <= 129 && pad_left >= 0 && pad_left < Kw && x && w && y, "cvft_dwconv1d_fwd: bad args");
    dim3 grid(B * ((T + 63) / 64), (C + 63) / 64);
    size_t sm = (size_t)(64 + Kw - 1) * 64 * sizeof(float);
    if (dtype == CVFT_F32)
        hipLaunchKernelGGL((dwconv_kernel<float, false>), grid, dim3(256), sm, (hipStream_t)stream, B, T, C, Kw, pad_left, (const float*)x, w, bias, len, (float*)y);
    else
        hipLaunchKernelGGL((dwconv_kernel<bf16_t, false>), grid, dim3(256), sm, (hipStream_t)stream, B, T, C, Kw, pad_left, (const bf16_t*)x, w, bias, len, (bf16_t*)y);
    CVFT_LAUNCH_CHECK("cvft_dwconv1d_fwd");
    return 0;
}
extern "C" int cvft_dwconv1d_bwd(int dtype, int B, int T, int C, int Kw, int pad_left, const void* dy, const float* w,
                                 const int32_t* len, void* dx, void* stream) {
    CHECK_DTYPE("cvft_dwconv1d_bwd", dtype);
    CVFT_CHECK_ARG(B > 0 && T > 0 && C > 0 && Kw > 0 && Kw <= 129 && pad_left >= 0 && pad_left < Kw && dy && w && dx, "cvft_dwconv1d_bwd: bad args");
    dim3 grid(B * ((T + 63) / 64), (C + 63) / 64);
    size_t sm = (size_t)(64 + Kw - 1) * 64 * sizeof(float);
    if (dtype == CVFT_F32)
        hipLaunchKernelGGL((dwconv_kernel<float, true>), grid, dim3(256), sm, (hipStream_t)stream, B, T, C, Kw, pad_left, (const float*)dy, w, nullptr, len, (float*)dx);
    else
        hipLaunchKernelGGL((dwconv_kernel<bf16_t, true>), grid, dim3(256), sm, (hipStream_t)stream, B, T, C, Kw, pad_left, (const bf16_t*)dy, w, nullptr, len, (bf16_t*)dx);
    CVFT_LAUNCH_CHECK("cvft_dwconv1d_bwd");
    return 0;
}

// ---------------------------------------------------------------- flat optimiser pieces
__global__ void __launch_bounds__(256) sumsq_kernel(size_t n, const float* __restrict__ g, float* __restrict__ out) {
    __shared__ float sm[16];
    float s = 0.f;
    EW_LOOP(i, n) s += g[i] * g[i];
    s = block_sum(s, sm);
    if (threadIdx.x == 0) atomicAdd(out, s);
}
// Order-fixed form: per-block partials written to a caller buffer, then ONE block adds them in index order.
// The gradient norm -- hence the clip coefficient and the AdamW update -- is bitwise reproducible run to run and
// identical on every data-parallel replica (the atomic form above depends on the order blocks finish in).
__global__ void __launch_bounds__(256) sumsq_part_kernel(size_t n, const float* __restrict__ g, float* __restrict__ part) {
    __shared__ float sm[16];
    float s = 0.f;
    EW_LOOP(i, n) s += g[i] * g[i];
    s = block_sum(s, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void __launch_bounds__(256) sumsq_final_kernel(int nparts, const float* __restrict__ part, float* __restrict__ out) {
    __shared__ float sm[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];       // fixed assignment, fixed order
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sm[0];
}
extern "C" int cvft_sumsq_ordered(int64_t n, const float* g, float* partials, float* out, void* stream) {
    CVFT_CHECK_ARG(n >= 0 && g && out && partials, "cvft_sumsq_ordered: bad args");
    unsigned grid = n > 0 ? ew_grid((size_t)n) : 1;
    if (grid > CVFT_SUMSQ_PARTS) grid = CVFT_SUMSQ_PARTS;
    hipLaunchKernelGGL(sumsq_part_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (size_t)n, g, partials);
    CVFT_LAUNCH_CHECK("cvft_sumsq_ordered");
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (int)grid, partials, out);
    CVFT_LAUNCH_CHECK("cvft_sumsq_ordered");
    return 0;
}

extern "C" int cvft_sumsq(int64_t n, const float* g, float* out, void* stream) {
    CVFT_CHECK_ARG(n >= 0 && g && out, "cvft_sumsq: bad args");
    if (n == 0) return 0;
    unsigned grid = ew_grid((size_t)n);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (size_t)n, g, out);
    CVFT_LAUNCH_CHECK("cvft_sumsq");
    return 0;
}

__global__ void adamw_flat_kernel(size_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                  float* __restrict__ v, const float* __restrict__ lr_p, float b1, float b2, float eps,
                                  float wd, const float* __restrict__ step_p, const float* __restrict__ gnorm_sq,
                                  float max_norm, float grad_scale) {
    const float lr = lr_p[0];
    const float step = step_p[0];
    float clip = 1.f;
    if (max_norm > 0.f && gnorm_sq) {
        float tn = sqrtf(gnorm_sq[0]) * fabsf(grad_scale);
        clip = fminf(1.f, max_norm / (tn + 1e-6f));
    }
    const float gs = grad_scale * clip;
    const float bc1 = 1.f - powf(b1, step);
    const float bc2 = 1.f - powf(b2, step);
    const float step_size = lr / bc1;
    const float bc2s = sqrtf(bc2);
    EW_LOOP(i, n) {
        float gi = g[i] * gs;
        float pi = p[i] * (1.f - lr * wd);
        float mi = b1 * m[i] + (1.f - b1) * gi;
        float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        float denom = sqrtf(vi) / bc2s + eps;
        p[i] = pi - step_size * (mi / denom);
        m[i] = mi;
        v[i] = vi;
    }
}
extern "C" int cvft_adamw_flat(int64_t n, float* p, const float* g, float* m, float* v, const float* lr, float beta1,
                               float beta2, float eps, float wd, const float* step, const float* gnorm_sq, float max_norm,
                               float grad_scale, void* stream) {
    CVFT_CHECK_ARG(n >= 0 && p && g && m && v && lr && step, "cvft_adamw_flat: bad args");
    if (n == 0) return 0;
    hipLaunchKernelGGL(adamw_flat_kernel, dim3(ew_grid((size_t)n)), dim3(256), 0, (hipStream_t)stream, (size_t)n, p, g, m, v, lr,
                       beta1, beta2, eps, wd, step, gnorm_sq, max_norm, grad_scale);
    CVFT_LAUNCH_CHECK("cvft_adamw_flat");
    return 0;
}

__global__ void cast_bf16_kernel(size_t n, const float* __restrict__ s, bf16_t* __restrict__ d) {
    EW_LOOP(i, n) d[i] = (bf16_t)s[i];
}
extern "C" int cvft_cast_f32_to_bf16(int64_t n, const float* src, void* dst, void* stream) {
    CVFT_CHECK_ARG(n >= 0 && src && dst, "cvft_cast_f32_to_bf16: bad args");
    if (n == 0) return 0;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(ew_grid((size_t)n)), dim3(256), 0, (hipStream_t)stream, (size_t)n, src, (bf16_t*)dst);
    CVFT_LAUNCH_CHECK("cvft_cast_f32_to_bf16");
    return 0;
}


// ------------------------------------------------------------------------------
// Inverted dropout with an optional residual add:  y = residual + keep(x) / (1 - p).
// (nn.Dropout call sites of the encoders: subsampling.py:84, embedding.py:285-288, encoder_layer.py:95-104 / 205-234,
//  positionwise_feed_forward.py:54.)  The keep mask is a pure function of (*seed, site, element index) -- SplitMix64
// finaliser on a 64-bit counter, 4 x 16 random bits per group of 4 elements -- so backward re-derives it instead of
// storing a mask tensor, and a captured hipGraph gets fresh masks on every replay because *seed lives on the device.
// Not torch's Philox stream: equality with the reference is statistical (keep rate, scale), as for any RNG change.
// ------------------------------------------------------------------------------
// MODE 0: y = res + drop(x);  MODE 1: y = drop(act(x))  (forward of act -> dropout);  MODE 2: y = drop(g) * act'(x), g = `res`
// (their backward in one pass).  VEC: n % 4 == 0 and 8-/16-byte aligned pointers -> one vector access per 4-element group.
template <typename T, int MODE, bool VEC>
__global__ void dropout_add_kernel(size_t n, const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y, float p,
                                   const long long* __restrict__ seed, unsigned site, int act) {
    const unsigned long long key = cvft_drop_key(seed, site);
    const unsigned thr = cvft_drop_thr(p);                                    // keep when the element's 16-bit field >= thr
    const float scale = 1.f / (1.f - p);
    const size_t n4 = (n + 3) / 4;
    typedef typename std::conditional<sizeof(T) == 2, uint2, uint4>::type V4;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < n4; g += (size_t)gridDim.x * blockDim.x) {
        bool kp[4];
        cvft_keep4(key, g, thr, kp);
        T xv[4], rv[4], ov[4];
        if (VEC) {
            *reinterpret_cast<V4*>(xv) = *reinterpret_cast<const V4*>(x + 4 * g);
            if (res) *reinterpret_cast<V4*>(rv) = *reinterpret_cast<const V4*>(res + 4 * g);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * g + e < n) { xv[e] = x[4 * g + e]; if (res) rv[e] = res[4 * g + e]; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v;
            if (MODE == 0) {
                v = kp[e] ? to_f32(xv[e]) * scale : 0.f;
                if (res) v += to_f32(rv[e]);
            } else if (MODE == 1) {
                v = kp[e] ? act_apply(act, to_f32(xv[e])) * scale : 0.f;
            } else {
                v = kp[e] ? to_f32(rv[e]) * scale * act_grad(act, to_f32(xv[e])) : 0.f;
            }
            ov[e] = from_f32<T>(v);
        }
        if (VEC) {
            *reinterpret_cast<V4*>(y + 4 * g) = *reinterpret_cast<const V4*>(ov);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * g + e < n) y[4 * g + e] = ov[e];
        }
    }
}
template <int MODE>
static void dropout_launch(int dtype, int64_t n, const void* x, const void* res, void* y, float p, const int64_t* seed, unsigned site,
                           int act, hipStream_t st) {
    const size_t n4 = ((size_t)n + 3) / 4;
    const uintptr_t al = (dtype == CVFT_F32) ? 15 : 7;
    const bool vec = (n % 4 == 0) && ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)res) & al) == 0);
#define DR_LAUNCH(TT, VV) hipLaunchKernelGGL((dropout_add_kernel<TT, MODE, VV>), dim3(ew_grid(n4)), dim3(256), 0, st, (size_t)n, (const TT*)x, \
                                             (const TT*)res, (TT*)y, p, (const long long*)seed, site, act)
    if (dtype == CVFT_F32) { if (vec) DR_LAUNCH(float, true); else DR_LAUNCH(float, false); }
    else { if (vec) DR_LAUNCH(bf16_t, true); else DR_LAUNCH(bf16_t, false); }
#undef DR_LAUNCH
}
extern "C" int cvft_dropout_add(int dtype, int64_t n, const void* x, const void* residual, void* y, float p,
                                const int64_t* seed, unsigned site, void* stream) {
    CHECK_DTYPE("cvft_dropout_add", dtype);
    CVFT_CHECK_ARG(n >= 0 && x && y && seed && cvft_drop_rate_ok(p), "cvft_dropout_add: bad args (p == 0 or 2^-16 <= p <= 1 - 2^-16)");
    if (n == 0) return 0;
    dropout_launch<0>(dtype, n, x, residual, y, p, seed, site, 0, (hipStream_t)stream);
    CVFT_LAUNCH_CHECK("cvft_dropout_add");
    return 0;
}
// h = dropout(act(z)) (positionwise_feed_forward.py:54) and its backward dz = keep/(1-p) * dh * act'(z), one pass each
// (dh == NULL: forward).
extern "C" int cvft_act_dropout(int dtype, int64_t n, int act, const void* z, const void* dh, void* y, float p,
                                const int64_t* seed, unsigned site, void* stream) {
    CHECK_DTYPE("cvft_act_dropout", dtype);
    CVFT_CHECK_ARG(n >= 0 && z && y && seed && cvft_drop_rate_ok(p), "cvft_act_dropout: bad args (p == 0 or 2^-16 <= p <= 1 - 2^-16)");
    if (n == 0) return 0;
    if (dh) dropout_launch<2>(dtype, n, z, dh, y, p, seed, site, act, (hipStream_t)stream);
    else dropout_launch<1>(dtype, n, z, nullptr, y, p, seed, site, act, (hipStream_t)stream);
    CVFT_LAUNCH_CHECK("cvft_act_dropout");
    return 0;
}


// ------------------------------------------------------------------------------
// LoRA side dgrad under lora_dropout (lora.py:70-73 backward):  the side path saw drop(x), so its input gradient is
//   out[m,k] = dx[m,k] + sum_t keep_t(m*K + k)/(1-p) * sum_{j<16} V[m,16t+j] * A[16t+j][k]
// (t = adapters stacked on this input: 1, or 3 for q|k|v, each with its own mask site).  Memory-bound: one pass over
// [M,K]; A slice and V rows staged in LDS; masks from the shared counter-based generator.
// ------------------------------------------------------------------------------
template <int RT>
__global__ void __launch_bounds__(256) lora_side_dgrad_kernel(int M, int K, const bf16_t* __restrict__ V, int ldv,
                                                              const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ dx,
                                                              int ldi, bf16_t* __restrict__ out, int ldo, float p,
                                                              const long long* __restrict__ seed, uint4 sites) {
    constexpr int R = 16 * RT;
    __shared__ float As[R][64 + 1];
    __shared__ float Vs[64][R + 1];
    const int k0 = blockIdx.x * 64, m0 = blockIdx.y * 64, tid = threadIdx.x;
    for (int e = tid; e < R * 64; e += 256) {
        const int j = e / 64, c = e % 64;
        As[j][c] = (k0 + c < K) ? to_f32(A[(size_t)j * lda + k0 + c]) : 0.f;
    }
    for (int e = tid; e < 64 * R; e += 256) {
        const int r = e / R, j = e % R;
        Vs[r][j] = (m0 + r < M) ? to_f32(V[(size_t)(m0 + r) * ldv + j]) : 0.f;
    }
    __syncthreads();
    const unsigned thr = cvft_drop_thr(p);
    const float scale = 1.f / (1.f - p);
    const unsigned st[4] = {sites.x, sites.y, sites.z, sites.w};
    unsigned long long keys[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) keys[t] = cvft_drop_key(seed, st[t]);
    const int cg = tid & 15, rr = tid >> 4;                       // 16 column groups of 4 x 16 row slots
    const int kk = k0 + cg * 4;
    if (kk >= K) return;
    for (int i = 0; i < 4; ++i) {
        const int r = rr + 16 * i, m = m0 + r;
        if (m >= M) break;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            bool kp[4];
            cvft_keep4(keys[t], ((unsigned long long)m * K + kk) >> 2, thr, kp);
            float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float v = Vs[r][16 * t + j];
#pragma unroll
                for (int e = 0; e < 4; ++e) part[e] += v * As[16 * t + j][cg * 4 + e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += kp[e] ? part[e] * scale : 0.f;
        }
        const bf16x4 d = *reinterpret_cast<const bf16x4*>(dx + (size_t)m * ldi + kk);
        bf16x4 o = {(bf16_t)((float)d[0] + acc[0]), (bf16_t)((float)d[1] + acc[1]), (bf16_t)((float)d[2] + acc[2]), (bf16_t)((float)d[3] + acc[3])};
        *reinterpret_cast<bf16x4*>(out + (size_t)m * ldo + kk) = o;
    }
}
extern "C" int cvft_lora_side_dgrad(int M, int K, int R, const void* V, int ldv, const void* A, int lda, const void* dx, int ldi,
                                    void* out, int ldo, float p, const int64_t* seed, const unsigned* sites, void* stream) {
    CVFT_CHECK_ARG(M > 0 && K > 0 && K % 4 == 0 && (R == 16 || R == 48) && V && A && dx && out && seed && sites && ldv >= R && lda >= K &&
                   ldi >= K && ldo >= K && ldi % 4 == 0 && ldo % 4 == 0 && p > 0.f && cvft_drop_rate_ok(p) &&
                   (((uintptr_t)dx | (uintptr_t)out) & 7) == 0, "cvft_lora_side_dgrad: bad args (bf16, K %% 4 == 0, R in {16, 48})");
    dim3 grid((K + 63) / 64, (M + 63) / 64);
    uint4 st = make_uint4(sites[0], R > 16 ? sites[1] : 0u, R > 16 ? sites[2] : 0u, 0u);
    if (R == 16)
        hipLaunchKernelGGL((lora_side_dgrad_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, M, K, (const bf16_t*)V, ldv, (const bf16_t*)A, lda,
                           (const bf16_t*)dx, ldi, (bf16_t*)out, ldo, p, (const long long*)seed, st);
    else
        hipLaunchKernelGGL((lora_side_dgrad_kernel<3>), grid, dim3(256), 0, (hipStream_t)stream, M, K, (const bf16_t*)V, ldv, (const bf16_t*)A, lda,
                           (const bf16_t*)dx, ldi, (bf16_t*)out, ldo, p, (const long long*)seed, st);
    CVFT_LAUNCH_CHECK("cvft_lora_side_dgrad");
    return 0;
}
