// Micro-benchmark (tools only): can ONE wave per SIMD overlap its own VALU work with its own MFMAs?  Per loop step: 4 independent
// v_mfma_f32_32x32x16_bf16 (128 cycles of matrix pipe) and NV independent VALU instructions (fma / packed fma / exp+rcp), fenced per
// step.  If they overlap, cycles per step stay ~128 until the VALU work exceeds it; if not, they add up.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// KIND 0: scalar fma chain x NV; 1: packed fma x NV; 2: NV/2 x (exp2 + rcp)
template <int NM, int NV, int KIND>
__global__ __launch_bounds__(512, 1) void k(int n, float* out, unsigned long long* stamps) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
    for (int r = 0; r < 4; ++r) for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
    float v[8];
    f32x2 pv[8];
    for (int i = 0; i < 8; ++i) { v[i] = 0.001f * lane + i; pv[i] = f32x2{v[i], v[i] + 1.f}; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int r = 0; r < NM; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[r], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (KIND == 0) v[j & 7] = fmaf(v[j & 7], 1.0001f, 0.5f);
            if (KIND == 1) pv[j & 7] = pv[j & 7] * f32x2{1.0001f, 1.0002f} + f32x2{0.5f, 0.25f};
            if (KIND == 2) v[j & 7] = (j & 1) ? __builtin_amdgcn_rcpf(v[j & 7]) : __builtin_amdgcn_exp2f(v[j & 7]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int r = 0; r < 4; ++r) for (int i = 0; i < 16; ++i) s += acc[r][i];
    for (int i = 0; i < 8; ++i) s += v[i] + pv[i][0] + pv[i][1];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[0] = t1 - t0;
    if (s == 123.456f) out[threadIdx.x] = s;
}

int main() {
    float* out; unsigned long long* st; CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&st, 64));
    const int n = 2048;
    unsigned long long h;
#define RUN(NM, NV, KIND) RUNB(NM, NV, KIND, 256) RUNB(NM, NV, KIND, 512)
#define RUNB(NM, NV, KIND, TPB) { for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<NM, NV, KIND>), dim3(64), dim3(TPB), 0, 0, n, out, st); CK(hipDeviceSynchronize()); \
        CK(hipMemcpy(&h, st, 8, hipMemcpyDeviceToHost)); printf("%d waves/SIMD: MFMA x%d + VALU x%2d kind %d: %7.1f cycles per step\n", TPB / 256, NM, NV, KIND, (double)h / n); }
    RUN(4, 0, 0) RUN(0, 16, 0) RUN(0, 32, 0) RUN(4, 16, 0) RUN(4, 32, 0) RUN(4, 48, 0)
    RUN(0, 32, 1) RUN(4, 16, 1) RUN(4, 32, 1) RUN(4, 48, 1)
    RUN(0, 16, 2) RUN(4, 8, 2) RUN(4, 16, 2)
    RUN(2, 0, 0) RUN(2, 16, 0) RUN(2, 32, 0)
    return 0;
}
