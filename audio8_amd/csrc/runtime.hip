// a8hip runtime glue: version, thread-local error text, launch checking.
#include "a8_common.h"
#include "a8hip_internal.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

int a8_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int a8_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return a8_set_error(A8_ERR_LAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
    return A8_OK;
}

extern "C" int a8_version(void) { return A8HIP_VERSION; }
extern "C" const char* a8_last_error(void) { return g_err; }

int g_a8_tuning[32] = {0, 802, 0, 41, 41, 20000, 3, 0, 0, 0, 0, 0, 0, 0, 1000, 1, 0, 801, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

extern "C" int a8_set_tuning(int key, int value) {
    if (key < 0 || key >= 32) return a8_set_error(A8_ERR_ARG, "a8_set_tuning: unknown key %d", key);
    g_a8_tuning[key] = value;
    return A8_OK;
}
