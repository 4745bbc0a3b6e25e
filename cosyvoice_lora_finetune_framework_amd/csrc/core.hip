// core.hip -- version + thread-local error string of libcvft.
#include <stdarg.h>
#include "common.cuh"

static thread_local char g_err[512] = "";

extern "C" void cvft_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* cvft_last_error(void) { return g_err; }
extern "C" int cvft_version(void) { return 100; }
