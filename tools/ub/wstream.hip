// Micro-benchmark (tools only, not product): how fast can ONE workgroup per CU stream pre-packed MFMA weight fragments
// (1 KB per wave-instruction, L2-resident 1 MB table shared by all workgroups) while issuing RT MFMAs per fragment?
//   mode 0: global_load_dwordx4 straight to VGPRs, PF fragments in flight per wave
//   mode 1: LDS-DMA (global_load_lds_dwordx4) into a per-wave ring of PF 1-KB slots, ds_read_b128, MFMA
// Sizes the fused estimator feed-forward kernel (block_fused.hip): DESIGN section 7.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int PF, int RT>
__global__ __launch_bounds__(256, 1) void k_reg(const bf16x8* __restrict__ W, int nfrag_per_wave, float* out, int wbytes_mask) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bf16x8* p = W + (size_t)wave * nfrag_per_wave * 64 + lane;
    f32x16 acc[RT];
    bf16x8 y[RT];
    for (int r = 0; r < RT; ++r) { for (int i = 0; i < 16; ++i) acc[r][i] = 0.f; for (int i = 0; i < 8; ++i) y[r][i] = (__bf16)(0.01f * (lane + i + r)); }
    constexpr int H = PF / 2;
    bf16x8 bufA[H], bufB[H];
#pragma unroll
    for (int j = 0; j < H; ++j) bufA[j] = p[(size_t)j * 64];
    p += (size_t)H * 64;
    // two named register sets: the loads of one set are issued in front of the MFMAs that consume the other (reads run past the
    // table's end by PF fragments: the table is over-allocated)
    for (int i = 0; i < nfrag_per_wave; i += PF) {
#pragma unroll
        for (int j = 0; j < H; ++j) bufB[j] = p[(size_t)j * 64];
        p += (size_t)H * 64;
#pragma unroll
        for (int j = 0; j < H; ++j)
#pragma unroll
            for (int r = 0; r < RT; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bufA[j], y[r], acc[r], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < H; ++j) bufA[j] = p[(size_t)j * 64];
        p += (size_t)H * 64;
#pragma unroll
        for (int j = 0; j < H; ++j)
#pragma unroll
            for (int r = 0; r < RT; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bufB[j], y[r], acc[r], 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < RT; ++r) for (int i = 0; i < 16; ++i) s += acc[r][i];
    if (s == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int PF, int RT>
__global__ __launch_bounds__(256, 1) void k_lds(const bf16x8* __restrict__ W, int nfrag_per_wave, float* out, int) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bf16x8* p = W + (size_t)wave * nfrag_per_wave * 64 + lane;
    char* ring = smem + wave * PF * 1024;
    f32x16 acc[RT];
    bf16x8 y[RT];
    for (int r = 0; r < RT; ++r) { for (int i = 0; i < 16; ++i) acc[r][i] = 0.f; for (int i = 0; i < 8; ++i) y[r][i] = (__bf16)(0.01f * (lane + i + r)); }
#pragma unroll
    for (int j = 0; j < PF; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (size_t)j * 64),
                                         (__attribute__((address_space(3))) void*)(ring + j * 1024), 16, 0, 0);
    for (int i = 0; i < nfrag_per_wave; i += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            // oldest outstanding DMA = slot j
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PF - 1) : "memory");
            bf16x8 f = *reinterpret_cast<const bf16x8*>(ring + j * 1024 + lane * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            int nx = i + PF + j; nx = nx < nfrag_per_wave ? nx : nfrag_per_wave - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (size_t)nx * 64),
                                             (__attribute__((address_space(3))) void*)(ring + j * 1024), 16, 0, 0);
#pragma unroll
            for (int r = 0; r < RT; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, y[r], acc[r], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int r = 0; r < RT; ++r) for (int i = 0; i < 16; ++i) s += acc[r][i];
    if (s == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <typename F>
static float timeit(F f, int reps = 2000) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 500; ++i) f();
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.f / reps;
}

int main() {
    const int nfrag_per_wave = 256;                     // 256 KB per wave, 1 MB per workgroup
    const size_t bytes = (size_t)4 * nfrag_per_wave * 1024;
    bf16x8* W; float* out;
    CK(hipMalloc(&W, bytes + (1 << 20))); CK(hipMalloc(&out, 1 << 22));
    std::vector<unsigned short> h(bytes / 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (rand() & 0x1ff) + ((rand() & 1) << 15);
    CK(hipMemcpy(W, h.data(), bytes, hipMemcpyHostToDevice));
    const int grids[] = {32, 63, 125, 250, 500};
    printf("mode PF RT grid us  GB/s_per_WG  TB/s_total\n");
#define RUN(KN, MODE, PF, RT, SH)                                                                        \
    for (int g : grids) {                                                                                 \
        if (SH > 65536) CK(hipFuncSetAttribute((const void*)KN<PF, RT>, hipFuncAttributeMaxDynamicSharedMemorySize, SH)); \
        float us = timeit([&] { hipLaunchKernelGGL((KN<PF, RT>), dim3(g), dim3(256), SH, 0, W, nfrag_per_wave, out, 0); }); \
        CK(hipGetLastError());                                                                            \
        printf("%d %2d %d %4d %7.2f %7.1f %6.2f\n", MODE, PF, RT, g, us, bytes / us / 1e3, g * (double)bytes / us / 1e6); \
    }
    RUN(k_reg, 0, 8, 1, 0)
    RUN(k_reg, 0, 16, 1, 0)
    RUN(k_reg, 0, 32, 1, 0)
    RUN(k_reg, 0, 16, 2, 0)
    RUN(k_reg, 0, 32, 2, 0)
    RUN(k_lds, 1, 8, 1, 4 * 8 * 1024)
    RUN(k_lds, 1, 16, 1, 4 * 16 * 1024)
    RUN(k_lds, 1, 32, 1, 4 * 32 * 1024)
    RUN(k_lds, 1, 32, 2, 4 * 32 * 1024)
    return 0;
}
