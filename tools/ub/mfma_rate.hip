// Micro-benchmark (tools only): cycles per v_mfma_f32_32x32x16_bf16 on one wave per SIMD (dependent chain vs NACC independent
// accumulators), the in-kernel clock, and the pure fragment-streaming rate of one workgroup per CU with NO arithmetic.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NACC>
__global__ __launch_bounds__(256, 1) void k_mfma(int n, float* out, unsigned long long* stamps) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[NACC];
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
    for (int r = 0; r < NACC; ++r) for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; i += NACC) {
#pragma unroll
        for (int r = 0; r < NACC; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[r], 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < NACC; ++r) for (int i = 0; i < 16; ++i) s += acc[r][i];
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
    if (s == 123.456f) out[threadIdx.x] = s;
}

// loads only: every fragment is xor-ed into one register (keeps the loads, no MFMA)
template <int PF>
__global__ __launch_bounds__(256, 1) void k_stream(const bf16x8* __restrict__ W, int nfrag, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint4* p = reinterpret_cast<const uint4*>(W) + (size_t)wave * nfrag * 64 + lane;
    uint4 x = {0, 0, 0, 0};
    for (int i = 0; i < nfrag; i += PF) {
        uint4 b[PF];
#pragma unroll
        for (int j = 0; j < PF; ++j) b[j] = p[(size_t)(i + j) * 64];
#pragma unroll
        for (int j = 0; j < PF; ++j) { x.x ^= b[j].x; x.y ^= b[j].y; x.z ^= b[j].z; x.w ^= b[j].w; }
    }
    if ((x.x ^ x.y ^ x.z ^ x.w) == 0x12345u) out[threadIdx.x] = 1.f;
}

template <typename F>
static float timeit(F f, int reps = 500) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 200; ++i) f();
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.f / reps;
}

int main() {
    float* out; unsigned long long* st; CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&st, 64));
    const int n = 4096;
    unsigned long long h[2];
#define RUNM(NACC, G) { float us = timeit([&] { hipLaunchKernelGGL((k_mfma<NACC>), dim3(G), dim3(256), 0, 0, n, out, st); }); \
        CK(hipMemcpy(h, st, 16, hipMemcpyDeviceToHost)); \
        printf("mfma NACC=%d grid=%3d: %7.2f us  %6.1f ns/mfma  in-kernel %6.1f cyc/mfma  clock %.2f GHz\n", NACC, G, us, us * 1e3 / n, (double)h[0] / n, (double)h[0] / (double)h[1] * 0.1); }
    RUNM(1, 32) RUNM(1, 256) RUNM(2, 32) RUNM(4, 32) RUNM(4, 256) RUNM(8, 256)
    const int nfrag = 256; const size_t bytes = (size_t)4 * nfrag * 1024;
    bf16x8* W; CK(hipMalloc(&W, bytes + (1 << 20))); CK(hipMemset(W, 1, bytes));
#define RUNS(PF, G) { float us = timeit([&] { hipLaunchKernelGGL((k_stream<PF>), dim3(G), dim3(256), 0, 0, W, nfrag, out); }); \
        printf("stream PF=%2d grid=%3d: %7.2f us  %6.1f GB/s per WG  %6.2f TB/s\n", PF, G, us, bytes / us / 1e3, G * (double)bytes / us / 1e6); }
    RUNS(8, 32) RUNS(8, 125) RUNS(8, 250) RUNS(16, 32) RUNS(16, 125) RUNS(16, 250) RUNS(32, 125) RUNS(32, 250) RUNS(64, 125) RUNS(64, 250)
    return 0;
}
