// a8hip runtime glue: version, thread-local error text, launch checking.
#include "a8_common.h"
#include "a8hip_internal.h"
#include <stdarg.h>
#include <stdio.h>
#include <atomic>

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

// Developer knobs (include/a8hip.h: a8_set_tuning).  PROCESS-WIDE by design -- they select among kernel variants whose results are
// bit-identical (asserted by tests/test_gemm_gpu.py and friends), never semantics, so two models or two threads in one process can only
// change each other's speed.  Every access is an atomic load / store: a launch sees either the old or the new value of a key,
// never a torn one; a change applies to launches ISSUED after a8_set_tuning returns.
A8TuningTable g_a8_tuning = {{0, 802, 0, 41, 41, 20000, 3, 0, 0, 0, 0, 0, 0, 0, 1000, 1, 0, 801, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};

extern "C" int a8_set_tuning(int key, int value) {
    if (key < 0 || key >= 32) return a8_set_error(A8_ERR_ARG, "a8_set_tuning: unknown key %d", key);
    g_a8_tuning.set(key, value);
    return A8_OK;
}
extern "C" int a8_get_tuning(int key, int* value) {
    if (key < 0 || key >= 32 || value == nullptr) return a8_set_error(A8_ERR_ARG, "a8_get_tuning: unknown key %d or null result", key);
    *value = g_a8_tuning[key];
    return A8_OK;
}

// One wave that holds its stream for `us` microseconds of the 100 MHz wall clock (s_memrealtime): the transport time of a collective in
// single-GPU exposure measurements (tools/exchange_exposure.py, GradReducer(fake_gbps=)).  It occupies one wave slot and no memory
// bandwidth -- unlike RCCL's kernels, which hold CUs: what it bounds is the un-overlapped part of the exchange, not its interference.
namespace {
__global__ __launch_bounds__(64) void delay_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
}  // namespace
extern "C" int a8_delay_us(float us, void* stream) {
    A8_REQUIRE(us >= 0.f && us < 1.0e6f, "a8_delay_us: %f us is out of range [0, 1 s)", (double)us);
    if (us == 0.f) return A8_OK;
    hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)(us * 100.0f));
    return a8_check_launch("a8_delay_us");
}
