// a8hip: conv layer 0 (Conv1d 1->512, k=10, s=5, no bias) + GroupNorm(512 groups = per-(b,c) instance norm
// over time) + exact GELU, forward and backward.   Reference: audio8/wav2vec2.py:417-423, 448-456.
//
// HBM-first design: the layer is 10 MACs per output, so instead of writing the pre-norm tensor
// (204.8 M elements at 8 x 250k samples) and re-reading it twice, every pass RECOMPUTES the conv from
// the waveform (1 MB per utterance, LDS-staged, coalesced along the sample axis):
//   forward  : stats pass (no output)  ->  finalize (mean, rstd per (b,c), f64)  ->  apply pass writes
//              gelu(gn(conv(x))) once, channels-last [B, L0, 512], 16 B per lane.
//   backward : reduce pass (sum dU, sum dU*xhat per (b,c))  ->  weight-grad pass; both read only dout.
// Lane = 8 consecutive channels, wave = a stripe of frames; the frame's 10 samples are a wave-uniform
// (broadcast) LDS read; per-(b,c) sums are kept in registers and combined across the 4 waves in LDS.
#include "a8_common.h"
#include "a8hip_internal.h"

namespace {

constexpr int KW = 10, ST = 5, CC = 512;

template <typename T> A8_DEV void ld8(const T* p, float (&v)[8]);
template <> A8_DEV void ld8<bf16>(const bf16* p, float (&v)[8]) {
    const bf16x8 x = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)x[i];
}
template <> A8_DEV void ld8<float>(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[i + 4] = b[i]; }
}
template <typename T> A8_DEV void st8(T* p, const float (&v)[8]);
template <> A8_DEV void st8<bf16>(bf16* p, const float (&v)[8]) {
    bf16x8 x;
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (bf16)v[i];
    *reinterpret_cast<bf16x8*>(p) = x;
}
template <> A8_DEV void st8<float>(float* p, const float (&v)[8]) {
    *reinterpret_cast<f32x4*>(p) = (f32x4){v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = (f32x4){v[4], v[5], v[6], v[7]};
}

// stage the samples a chunk of CH frames needs: x[b, 5*t0 .. 5*t0 + 5*CH + 5)
template <int CH> A8_DEV void stage_x(float* xs, const float* __restrict__ xb, int t0, int L) {
    const int s0 = ST * t0;
    for (int i = threadIdx.x; i < ST * CH + (KW - ST); i += 256) xs[i] = (s0 + i < L) ? xb[s0 + i] : 0.f;
}
A8_DEV void load_w(float (&w)[8][KW], const float* __restrict__ wg, int c0) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int k = 0; k < KW; ++k) w[j][k] = wg[(c0 + j) * KW + k];
}
A8_DEV void conv_frame(const float* xs, int tl, const float (&w)[8][KW], float (&v)[8], float (&xv)[KW]) {
#pragma unroll
    for (int k = 0; k < KW; ++k) xv[k] = xs[ST * tl + k];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < KW; ++k) a = fmaf(w[j][k], xv[k], a);
        v[j] = a;
    }
}

// ---- forward pass 1: per-chunk (sum, sumsq) per channel -> part[b][chunk][c][2]
constexpr int CHF = 256;
__global__ __launch_bounds__(256) void conv0_stats_kernel(const float* __restrict__ x, const float* __restrict__ wg,
                                                          float* __restrict__ part, int L, int L0, int nchunk) {
    __shared__ float xs[ST * CHF + 8];
    __shared__ float red[4][CC][2];
    const int b = blockIdx.y, chunk = blockIdx.x, t0 = chunk * CHF;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c0 = lane * 8;
    stage_x<CHF>(xs, x + (long long)b * L, t0, L);
    float w[8][KW];
    load_w(w, wg, c0);
    __syncthreads();
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
    for (int tl = wv; tl < CHF && t0 + tl < L0; tl += 4) {
        float v[8], xv[KW];
        conv_frame(xs, tl, w, v, xv);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[j] += v[j]; q[j] = fmaf(v[j], v[j], q[j]); }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[wv][c0 + j][0] = s[j]; red[wv][c0 + j][1] = q[j]; }
    __syncthreads();
    float* out = part + ((long long)b * nchunk + chunk) * CC * 2;
    for (int i = threadIdx.x; i < CC * 2; i += 256) {
        const int c = i >> 1, h = i & 1;
        out[i] = red[0][c][h] + red[1][c][h] + red[2][c][h] + red[3][c][h];
    }
}

// ---- forward pass 2: (mean, rstd) per (b,c) in f64
__global__ void conv0_finalize_kernel(const float* __restrict__ part, float* __restrict__ stats, int B, int nchunk, int L0, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * CC) return;
    const int b = i / CC, c = i % CC;
    double s = 0.0, q = 0.0;
    for (int ch = 0; ch < nchunk; ++ch) {
        const float* p = part + (((long long)b * nchunk + ch) * CC + c) * 2;
        s += (double)p[0];
        q += (double)p[1];
    }
    const double mu = s / L0;
    double var = q / L0 - mu * mu;
    if (var < 0.0) var = 0.0;
    stats[i * 2 + 0] = (float)mu;
    stats[i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// ---- forward pass 3: out = gelu(gn(conv(x)))
template <typename T>
__global__ __launch_bounds__(256) void conv0_apply_kernel(const float* __restrict__ x, const float* __restrict__ wg,
                                                          const float* __restrict__ gn_w, const float* __restrict__ gn_b,
                                                          const float* __restrict__ stats, T* __restrict__ out, int L, int L0) {
    __shared__ float xs[ST * CHF + 8];
    const int b = blockIdx.y, t0 = blockIdx.x * CHF;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c0 = lane * 8;
    stage_x<CHF>(xs, x + (long long)b * L, t0, L);
    float w[8][KW], sc[8], sh[8];
    load_w(w, wg, c0);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float mu = stats[((long long)b * CC + c0 + j) * 2], rs = stats[((long long)b * CC + c0 + j) * 2 + 1];
        sc[j] = rs * gn_w[c0 + j];
        sh[j] = gn_b[c0 + j] - mu * sc[j];
    }
    __syncthreads();
    for (int tl = wv; tl < CHF && t0 + tl < L0; tl += 4) {
        float v[8], xv[KW];
        conv_frame(xs, tl, w, v, xv);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = gelu_f(fmaf(v[j], sc[j], sh[j]));
        st8(out + ((long long)b * L0 + t0 + tl) * CC + c0, v);
    }
}

// ---- backward pass 1: red[b][c] = (sum_t dU, sum_t dU * xhat), dU = dout * gelu'(u)
constexpr int CHB = 1024;
template <typename T>
__global__ __launch_bounds__(256) void conv0_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ wg,
                                                               const float* __restrict__ gn_w, const float* __restrict__ gn_b,
                                                               const float* __restrict__ stats, const T* __restrict__ dout,
                                                               float* __restrict__ red, int L, int L0) {
    __shared__ float xs[ST * CHB + 8];
    __shared__ float rr[4][CC][2];
    const int b = blockIdx.y, t0 = blockIdx.x * CHB;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c0 = lane * 8;
    stage_x<CHB>(xs, x + (long long)b * L, t0, L);
    float w[8][KW], mu[8], rs[8], gw[8], gb[8];
    load_w(w, wg, c0);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        mu[j] = stats[((long long)b * CC + c0 + j) * 2];
        rs[j] = stats[((long long)b * CC + c0 + j) * 2 + 1];
        gw[j] = gn_w[c0 + j];
        gb[j] = gn_b[c0 + j];
    }
    __syncthreads();
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    for (int tl = wv; tl < CHB && t0 + tl < L0; tl += 4) {
        float v[8], xv[KW], g[8];
        conv_frame(xs, tl, w, v, xv);
        ld8(dout + ((long long)b * L0 + t0 + tl) * CC + c0, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = (v[j] - mu[j]) * rs[j];
            const float du = g[j] * gelu_grad_f(fmaf(xh, gw[j], gb[j]));
            s1[j] += du;
            s2[j] = fmaf(du, xh, s2[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { rr[wv][c0 + j][0] = s1[j]; rr[wv][c0 + j][1] = s2[j]; }
    __syncthreads();
    for (int i = threadIdx.x; i < CC * 2; i += 256) {
        const int c = i >> 1, h = i & 1;
        atomicAdd(red + ((long long)b * CC + c) * 2 + h, rr[0][c][h] + rr[1][c][h] + rr[2][c][h] + rr[3][c][h]);
    }
}

// ---- backward pass 1.5: dgn_b[c] += sum_b s1, dgn_w[c] += sum_b s2
__global__ void conv0_bwd_affine_kernel(const float* __restrict__ red, float* __restrict__ dgn_w, float* __restrict__ dgn_b, int B) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= CC) return;
    float a = 0.f, bsum = 0.f;
    for (int b = 0; b < B; ++b) { bsum += red[((long long)b * CC + c) * 2]; a += red[((long long)b * CC + c) * 2 + 1]; }
    dgn_w[c] += a;
    dgn_b[c] += bsum;
}

// ---- backward pass 2: dV = rstd*gamma*(dU - mean(dU) - xhat*mean(dU*xhat)); dw[c][k] += sum_t dV x[5t+k]
template <typename T>
__global__ __launch_bounds__(256) void conv0_bwd_dw_kernel(const float* __restrict__ x, const float* __restrict__ wg,
                                                           const float* __restrict__ gn_w, const float* __restrict__ gn_b,
                                                           const float* __restrict__ stats, const T* __restrict__ dout,
                                                           const float* __restrict__ red, float* __restrict__ dw, int L, int L0) {
    __shared__ float xs[ST * CHB + 8];
    __shared__ float acc_s[CC * KW];
    const int b = blockIdx.y, t0 = blockIdx.x * CHB;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c0 = lane * 8;
    stage_x<CHB>(xs, x + (long long)b * L, t0, L);
    for (int i = threadIdx.x; i < CC * KW; i += 256) acc_s[i] = 0.f;
    float w[8][KW], mu[8], rs[8], gw[8], gb[8], m1[8], m2[8];
    load_w(w, wg, c0);
    const float invL = 1.0f / (float)L0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        mu[j] = stats[((long long)b * CC + c0 + j) * 2];
        rs[j] = stats[((long long)b * CC + c0 + j) * 2 + 1];
        gw[j] = gn_w[c0 + j];
        gb[j] = gn_b[c0 + j];
        m1[j] = red[((long long)b * CC + c0 + j) * 2] * invL;
        m2[j] = red[((long long)b * CC + c0 + j) * 2 + 1] * invL;
    }
    __syncthreads();
    float a[8][KW];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int k = 0; k < KW; ++k) a[j][k] = 0.f;
    for (int tl = wv; tl < CHB && t0 + tl < L0; tl += 4) {
        float v[8], xv[KW], g[8];
        conv_frame(xs, tl, w, v, xv);
        ld8(dout + ((long long)b * L0 + t0 + tl) * CC + c0, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = (v[j] - mu[j]) * rs[j];
            const float du = g[j] * gelu_grad_f(fmaf(xh, gw[j], gb[j]));
            const float dv = rs[j] * gw[j] * (du - m1[j] - xh * m2[j]);
#pragma unroll
            for (int k = 0; k < KW; ++k) a[j][k] = fmaf(dv, xv[k], a[j][k]);
        }
    }
    for (int ww = 0; ww < 4; ++ww) {
        if (wv == ww) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int k = 0; k < KW; ++k) acc_s[(c0 + j) * KW + k] += a[j][k];
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < CC * KW; i += 256) atomicAdd(dw + i, acc_s[i]);
}

}  // namespace

static int conv0_out_len(int L) { return (L - KW) / ST + 1; }

extern "C" int a8_conv0_gn_gelu_fwd(int dtype, const float* x, const float* w, const float* gn_w, const float* gn_b,
                                    void* out, float* stats, float* part, int B, int L, int C_, float eps, void* stream) {
    A8_REQUIRE(C_ == CC, "a8_conv0_gn_gelu_fwd: C must be %d (got %d)", CC, C_);
    A8_REQUIRE(L >= KW && B >= 1, "a8_conv0_gn_gelu_fwd: bad B=%d L=%d", B, L);
    A8_REQUIRE(dtype == A8_F32 || dtype == A8_BF16, "a8_conv0_gn_gelu_fwd: bad dtype");
    const int L0 = conv0_out_len(L), nchunk = cdiv(L0, CHF);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(conv0_stats_kernel, dim3(nchunk, B), dim3(256), 0, s, x, w, part, L, L0, nchunk);
    hipLaunchKernelGGL(conv0_finalize_kernel, dim3(cdiv(B * CC, 256)), dim3(256), 0, s, part, stats, B, nchunk, L0, eps);
    if (dtype == A8_BF16)
        hipLaunchKernelGGL(conv0_apply_kernel<bf16>, dim3(nchunk, B), dim3(256), 0, s, x, w, gn_w, gn_b, stats, (bf16*)out, L, L0);
    else
        hipLaunchKernelGGL(conv0_apply_kernel<float>, dim3(nchunk, B), dim3(256), 0, s, x, w, gn_w, gn_b, stats, (float*)out, L, L0);
    return a8_check_launch("a8_conv0_gn_gelu_fwd");
}

extern "C" int a8_conv0_gn_gelu_bwd(int dtype, const float* x, const float* w, const float* gn_w, const float* gn_b,
                                    const float* stats, const void* dout, float* dw, float* dgn_w, float* dgn_b,
                                    float* red, int B, int L, int C_, void* stream) {
    A8_REQUIRE(C_ == CC, "a8_conv0_gn_gelu_bwd: C must be %d (got %d)", CC, C_);
    A8_REQUIRE(L >= KW && B >= 1, "a8_conv0_gn_gelu_bwd: bad B=%d L=%d", B, L);
    A8_REQUIRE(dtype == A8_F32 || dtype == A8_BF16, "a8_conv0_gn_gelu_bwd: bad dtype");
    const int L0 = conv0_out_len(L), nchunk = cdiv(L0, CHB);
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(red, 0, sizeof(float) * 2 * B * CC, s) != hipSuccess) return a8_set_error(A8_ERR_LAUNCH, "a8_conv0_gn_gelu_bwd: memset failed");
    if (dtype == A8_BF16) {
        hipLaunchKernelGGL(conv0_bwd_reduce_kernel<bf16>, dim3(nchunk, B), dim3(256), 0, s, x, w, gn_w, gn_b, stats, (const bf16*)dout, red, L, L0);
        hipLaunchKernelGGL(conv0_bwd_affine_kernel, dim3(2), dim3(256), 0, s, red, dgn_w, dgn_b, B);
        hipLaunchKernelGGL(conv0_bwd_dw_kernel<bf16>, dim3(nchunk, B), dim3(256), 0, s, x, w, gn_w, gn_b, stats, (const bf16*)dout, red, dw, L, L0);
    } else {
        hipLaunchKernelGGL(conv0_bwd_reduce_kernel<float>, dim3(nchunk, B), dim3(256), 0, s, x, w, gn_w, gn_b, stats, (const float*)dout, red, L, L0);
        hipLaunchKernelGGL(conv0_bwd_affine_kernel, dim3(2), dim3(256), 0, s, red, dgn_w, dgn_b, B);
        hipLaunchKernelGGL(conv0_bwd_dw_kernel<float>, dim3(nchunk, B), dim3(256), 0, s, x, w, gn_w, gn_b, stats, (const float*)dout, red, dw, L, L0);
    }
    return a8_check_launch("a8_conv0_gn_gelu_bwd");
}
