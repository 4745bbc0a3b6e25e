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
extern "C" int cvft_version(void) { return 100; }
