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
