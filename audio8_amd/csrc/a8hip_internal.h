// internal: pulls in the public C ABI next to the device helpers
#pragma once
#include "../../include/a8hip.h"
#include <atomic>

// a8_set_tuning table (runtime.hip): [1] ln_bwd geometry NW*100 + rows per wave, [2..4] attention fwd / dK,dV / dQ launch geometry ([3], [4] = 32: the 32 x 32-MFMA backward kernels),
// [5] min M for the stream NT GEMM of gemm_stream.hip (0 = off), [7] its forced tile height (0 = per launch), [8] its grid oversubscription, [6] posconv wgrad loop splits, [9] conv layer 0 on the VALU,
// [14] min M for the asymmetric-ring NT GEMM, [17] ln_bwd geometry of the partial-row form, [19] stream GEMM for plain problems too, [20] / [21] AdamW grid / cache policy,
// [22] = 1: LDS-staged epilogue in the NT kernels, [23] = 1: residual add in the FFN data gradient's epilogue, [25] = 1: predicate-form operand addressing; the rest spare
// process-wide, atomically read and written (runtime.hip); g_a8_tuning[k] reads key k as a plain int
struct A8TuningTable {
    std::atomic<int> v[32];
    int operator[](int k) const { return v[k].load(std::memory_order_relaxed); }
    void set(int k, int value) { v[k].store(value, std::memory_order_relaxed); }
};
extern A8TuningTable g_a8_tuning;

// gemm_stream.hip: launches the problem on BM x 256 tiles if it qualifies (1), 0 = not taken, < 0 = -(error code)
int a8i_gemm_stream_try(const a8_gemm_desc* d, void* stream);
bool a8i_gemm_stream_eligible(const a8_gemm_desc* d);

// forward entry points with a dropout-counter origin (used by composite.hip for calls on a slice of the batch)
int a8i_layernorm_fwd(int dtype, const void* x, const void* res, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                      int rows, int cols, float eps, float p, uint32_t seed, uint32_t rng_stream, int64_t rng_elem0, void* stream);
int a8i_attn_fwd(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* keymask, int B, int T, int H,
                 int64_t ld_qkv, int64_t ld_o, float scale, float p, uint32_t seed, int bh0, void* stream);
int a8i_layernorm_bwd(int dtype, const void* dy, const void* dy2, const void* x, const void* res, const float* gamma, const float* mean,
                      const float* rstd, void* dx, void* dres, float* dgamma, float* dbeta, float* dbias, int rows, int cols, float p,
                      uint32_t seed, uint32_t rng_stream, int64_t rng_elem0, void* stream, float* partials = nullptr);
int a8i_layernorm_bwd_partial_rows(int rows);
int a8i_attn_bwd(int dtype, const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                 const uint8_t* keymask, void* dq, void* dk, void* dv, float* delta, float* dbias, int B, int T, int H, int64_t ld_qkv,
                 int64_t ld_o, int64_t ld_dqkv, float scale, float p, uint32_t seed, int bh0, void* stream);
