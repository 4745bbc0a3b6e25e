// core.hip -- version + thread-local error string of libcvft.
#include <stdarg.h>
#include "common.h"

static thread_local char g_err[512] = "";

extern "C" void cvft_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* cvft_last_error(void) { return g_err; }
static thread_local char g_kernel[96] = "";
extern "C" void cvft_set_kernel_label(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
    va_end(ap);
}
extern "C" const char* cvft_gemm_last_kernel(void) { return g_kernel; }
extern "C" int cvft_version(void) { return 101; }

// Execution hint (performance only, never results): how many independent kernel chains the caller runs concurrently on
// separate streams.  Kernels that own a whole CU (one workgroup per CU: the 96x256 GEMM tile) cannot share it with another
// chain's workgroups, so with three chains in flight tile shapes that co-reside are preferred (gemm_glds.hip).
static int g_chains = 1;
extern "C" int cvft_set_concurrent_chains(int n) { const int old = g_chains; g_chains = n < 1 ? 1 : n; return old; }
extern "C" int cvft_concurrent_chains(void) { return g_chains; }

// Diagnostic: a one-thread kernel that writes the device's wall clock (s_memrealtime, 100 MHz) into buf[slot] -- launched between
// the kernels of a stream (also inside a captured hipGraph, where HIP refuses timing events) it says WHEN the stream got there
// (llm_flow_model.py, CVFT_CHAIN_EVENTS).
__global__ void debug_stamp_kernel(unsigned long long* buf, int slot) {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    buf[slot] = t;
}
extern "C" int cvft_debug_stamp(unsigned long long* buf, int slot, void* stream) {
    CVFT_CHECK_ARG(buf && slot >= 0, "cvft_debug_stamp: null buffer / negative slot");
    hipLaunchKernelGGL(debug_stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, buf, slot);
    CVFT_LAUNCH_CHECK("cvft_debug_stamp");
    return 0;
}
