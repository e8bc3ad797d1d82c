// a8hip: HBM-bound row kernels -- LayerNorm fwd/bwd (with fused residual add + dropout), GELU backward
// with bias reduction, mask fill / gather / scatter, col2im for the strided-conv dgrad, casts,
// log-softmax, grad-norm and AdamW.  All loads/stores are 16 bytes per lane along the channel axis;
// one wave owns a row, statistics and column sums stay in registers (wavefront shuffles), and column
// sums leave the block as one f32 atomic per column.
#include "a8_common.h"
#include <algorithm>
#include "a8hip_internal.h"

namespace {

template <typename T> A8_DEV void load8(const T* p, float (&v)[8]);
template <> A8_DEV void load8<bf16>(const bf16* p, float (&v)[8]) {
    const bf16x8 x = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)x[i];
}
template <> A8_DEV void load8<float>(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[i + 4] = b[i]; }
}
template <typename T> A8_DEV void store8(T* p, const float (&v)[8]);
template <> A8_DEV void store8<bf16>(bf16* p, const float (&v)[8]) {
    bf16x8 x;
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (bf16)v[i];
    *reinterpret_cast<bf16x8*>(p) = x;
}
template <> A8_DEV void store8<float>(float* p, const float (&v)[8]) {
    *reinterpret_cast<f32x4*>(p) = (f32x4){v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = (f32x4){v[4], v[5], v[6], v[7]};
}

constexpr int MAXC = 3;   // chunks of 8 columns per lane -> cols <= 1536

struct Drop { float scale; uint32_t thr, seed, stream; };     // thr = thr7 * 0x01010101 (0 = dropout off), see a8_keep8
static Drop make_drop(float p, uint32_t seed, uint32_t stream) {
    Drop d;
    uint32_t t7 = p > 0.f ? (uint32_t)(p * 128.0f + 0.5f) : 0u;       // the drop probability is realised as t7 / 128
    if (p > 0.f && t7 == 0) t7 = 1;
    if (t7 > 127u) t7 = 127u;
    d.scale = t7 ? 128.0f / (128.0f - (float)t7) : 1.0f;
    d.thr = t7 * 0x01010101u;
    d.seed = seed; d.stream = stream;
    return d;
}

// ------------------------------------------------------------------ LayerNorm forward
// MC = 16-byte chunks per lane (cols <= 512 * MC).  gamma / beta are row-independent: loaded once per wave ahead of the row loop
// (inside it they were one more dependent memory round trip between the variance reduction and the stores).
template <typename T, int MC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     T* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
                                                     int rows, int cols, float eps, Drop dr, long long rng0) {
    const int lane = threadIdx.x & 63, nch = cols >> 3;
    const float inv = 1.0f / (float)cols;
    float gm[MC][8], bt[MC][8];
#pragma unroll
    for (int c = 0; c < MC; ++c)
        if (lane + 64 * c < nch) {
            load8(gamma + (lane + 64 * c) * 8, gm[c]);
            load8(beta + (lane + 64 * c) * 8, bt[c]);
        }
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        float v[MC][8];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < MC; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                const long long off = (long long)row * cols + ch * 8;
                load8(x + off, v[c]);
                if (dr.thr) {
                    const Keep8 kp = a8_keep8(dr.seed, dr.stream, (uint32_t)((off + rng0) >> 3), dr.thr);     // rng0: see a8i_layernorm_fwd
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[c][i] = a8_kept(kp, i) ? v[c][i] * dr.scale : 0.f;
                }
                if (res) {
                    float r[8];
                    load8(res + off, r);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[c][i] += r[i];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) s += v[c][i];
            }
        }
        const float mu = wave_sum(s) * inv;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (lane + 64 * c < nch) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float d = v[c][i] - mu; q += d * d; }
            }
        const float rs = rsqrtf(wave_sum(q) * inv + eps);
#pragma unroll
        for (int c = 0; c < MC; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                float o[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = (v[c][i] - mu) * rs * gm[c][i] + bt[c][i];
                store8(y + (long long)row * cols + ch * 8, o);
            }
        }
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}

// Column sums held per wave in registers -> LDS [NS][cols] (one slab per wave; with more than NS waves the upper
// waves add into the lower waves' slabs in a second pass) -> one atomic per column per block.
// acc[c][i] belongs to column (lane + 64c)*8 + i.  lds: float[NS * cols], NS = min(NW, 8).
template <int NW, int MC>
A8_DEV void flush_colsum(float (&acc)[MC][8], float* lds, float* __restrict__ out, int cols, const bool store = false) {
    constexpr int NS = NW < 8 ? NW : 8;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nch = cols >> 3;
    __syncthreads();
    if (w < NS) {
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (lane + 64 * c < nch) {
                float* p = lds + w * cols + (lane + 64 * c) * 8;
                *reinterpret_cast<f32x4*>(p) = (f32x4){acc[c][0], acc[c][1], acc[c][2], acc[c][3]};
                *reinterpret_cast<f32x4*>(p + 4) = (f32x4){acc[c][4], acc[c][5], acc[c][6], acc[c][7]};
            }
    }
    __syncthreads();
    if constexpr (NW > NS) {
        if (w >= NS) {
#pragma unroll
            for (int c = 0; c < MC; ++c)
                if (lane + 64 * c < nch) {
                    float* p = lds + (w - NS) * cols + (lane + 64 * c) * 8;
                    const f32x4 a = *reinterpret_cast<f32x4*>(p), b = *reinterpret_cast<f32x4*>(p + 4);
                    *reinterpret_cast<f32x4*>(p) = (f32x4){a[0] + acc[c][0], a[1] + acc[c][1], a[2] + acc[c][2], a[3] + acc[c][3]};
                    *reinterpret_cast<f32x4*>(p + 4) = (f32x4){b[0] + acc[c][4], b[1] + acc[c][5], b[2] + acc[c][6], b[3] + acc[c][7]};
                }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < cols; i += NW * 64) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < NS; ++k) t += lds[k * cols + i];
        if (store) out[i] = t;                       // a per-block partial row (summed later, off the critical stream): no same-address atomics
        else atomicAdd(out + i, t);
    }
}

// ------------------------------------------------------------------ LayerNorm backward
// MC = 16-byte chunks per lane (cols <= 512 * MC).  dbias (optional) += column sums of the dx output, i.e. the bias
// gradient of the Linear whose output fed this LayerNorm through dropout (saves a separate column-sum pass).
template <typename T, int NW, int MC>
__global__ __launch_bounds__(NW * 64, MC <= 2 ? 4 : 2) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ dy2,
                                                         const T* __restrict__ x, const T* __restrict__ res,
                                                         const float* __restrict__ gamma, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, T* __restrict__ dx, T* __restrict__ dres,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dbias,
                                                         float* __restrict__ partials, int rows, int cols, Drop dr, long long rng0) {
    __shared__ __attribute__((aligned(16))) float lds[(NW < 8 ? NW : 8) * MC * 512];
    const int lane = threadIdx.x & 63, nch = cols >> 3;
    const float inv = 1.0f / (float)cols;
    float ag[MC][8], ab[MC][8], ad[MC][8];
#pragma unroll
    for (int c = 0; c < MC; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) { ag[c][i] = 0.f; ab[c][i] = 0.f; ad[c][i] = 0.f; }
    for (int row = blockIdx.x * NW + (threadIdx.x >> 6); row < rows; row += gridDim.x * NW) {
        const float mu = mean[row], rs = rstd[row];
        float xh[MC][8], g[MC][8];
        Keep8 keep[MC];                          // the dropout mask of this row's chunks: hashed once, used for x and for dx
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < MC; ++c) {
            const int ch = lane + 64 * c;
            keep[c] = Keep8{0u, 0u};
            if (ch < nch) {
                const long long off = (long long)row * cols + ch * 8;
                float v[8], gm[8];
                load8(x + off, v);
                if (dr.thr) {
                    keep[c] = a8_keep8(dr.seed, dr.stream, (uint32_t)((off + rng0) >> 3), dr.thr);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = a8_kept(keep[c], i) ? v[i] * dr.scale : 0.f;
                }
                if (res) {
                    float r[8];
                    load8(res + off, r);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] += r[i];
                }
                load8(dy + off, g[c]);
                if (dy2) {
                    float r[8];
                    load8(dy2 + off, r);
#pragma unroll
                    for (int i = 0; i < 8; ++i) g[c][i] += r[i];
                }
                load8(gamma + ch * 8, gm);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    xh[c][i] = (v[i] - mu) * rs;
                    ag[c][i] += g[c][i] * xh[c][i];
                    ab[c][i] += g[c][i];
                    g[c][i] *= gm[i];
                    s1 += g[c][i];
                    s2 += g[c][i] * xh[c][i];
                }
            }
        }
        const float a = wave_sum(s1) * inv, b = wave_sum(s2) * inv;
#pragma unroll
        for (int c = 0; c < MC; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                const long long off = (long long)row * cols + ch * 8;
                float o[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = rs * (g[c][i] - a - xh[c][i] * b);
                if (dres) store8(dres + off, o);
                if (dr.thr) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[i] = a8_kept(keep[c], i) ? o[i] * dr.scale : 0.f;
                }
                if (dx) store8(dx + off, o);
                if (dbias || partials) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) ad[c][i] += o[i];
                }
            }
        }
    }
    if (partials) {                                  // [block][dgamma | dbeta | dbias][cols]: plain coalesced stores (a8_colsum_partials_reduce adds them up)
        float* pr = partials + (long long)blockIdx.x * 3 * cols;
        flush_colsum<NW, MC>(ag, lds, pr, cols, true);
        flush_colsum<NW, MC>(ab, lds, pr + cols, cols, true);
        flush_colsum<NW, MC>(ad, lds, pr + 2 * cols, cols, true);
        return;
    }
    if (dgamma) {
        flush_colsum<NW, MC>(ag, lds, dgamma, cols);
        flush_colsum<NW, MC>(ab, lds, dbeta, cols);
    }
    if (dbias) flush_colsum<NW, MC>(ad, lds, dbias, cols);
}

// out_k[c] += sum over the partial rows b of ws[b][k][c]  (k = 0..2; a null out_k is skipped).  grid = (column chunks of 256, row stripes).
// Round 5: eight independent loads in flight per thread and up to 64 row stripes (it was one dependent-latency load per iteration over
// nrows / 16 rows: 37 us per launch for 7 MB, 0.53 ms per step of side-stream time).
__global__ __launch_bounds__(256) void colsum_partials_reduce_kernel(const float* __restrict__ ws, int nrows, int cols, float* __restrict__ o0,
                                                                      float* __restrict__ o1, float* __restrict__ o2) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= 3 * cols) return;
    float* const out = c < cols ? o0 : (c < 2 * cols ? o1 : o2);
    if (!out) return;
    const long long ld = 3ll * cols, step = (long long)gridDim.y * ld;
    const float* p = ws + (long long)blockIdx.y * ld + c;
    int b = blockIdx.y;
    float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (; b + 7 * (int)gridDim.y < nrows; b += 8 * gridDim.y, p += 8 * step) {
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] += p[u * step];
    }
    for (; b < nrows; b += gridDim.y, p += step) t[0] += *p;
    atomicAdd(out + (c % cols), ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7])));
}

// ------------------------------------------------------------------ dz = dy * gelu'(z); dbias += colsum(dz)
// column-tiled: blockIdx.x owns 512 columns (one 16-byte chunk per lane), blockIdx.y stripes the rows.
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_bias_kernel(const T* __restrict__ dy, const T* __restrict__ z, T* __restrict__ dz,
                                                            float* __restrict__ dbias, int rows, int cols) {
    __shared__ float lds[4][512];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = blockIdx.x * 512 + lane * 8;
    float ab[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (col < cols) {
        // four rows' loads in flight per wave (round 5: one dependent 16-byte load per iteration ran the pure column-sum use -- the q/k/v bias
        // gradient over dqkv -- at 1.25 TB/s, waves waiting 86 % of their cycles); sums stay in row order
        const int stride = gridDim.y * 4;
        int row = blockIdx.y * 4 + wv;
        for (; row + 3 * stride < rows; row += 4 * stride) {
            float g[4][8], zz[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) load8(dy + ((long long)(row + u * stride) * cols + col), g[u]);
            if (z) {
#pragma unroll
                for (int u = 0; u < 4; ++u) load8(z + ((long long)(row + u * stride) * cols + col), zz[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (z) mul_gelu_grad_n(g[u], zz[u]);
#pragma unroll
                for (int i = 0; i < 8; ++i) ab[i] += g[u][i];
                if (dz) store8(dz + ((long long)(row + u * stride) * cols + col), g[u]);
            }
        }
        for (; row < rows; row += stride) {
            const long long off = (long long)row * cols + col;
            float g[8];
            load8(dy + off, g);
            if (z) {
                float zz[8];
                load8(z + off, zz);
                mul_gelu_grad_n(g, zz);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) ab[i] += g[i];
            if (dz) store8(dz + off, g);
        }
    }
    if (!dbias) return;
#pragma unroll
    for (int i = 0; i < 8; ++i) lds[wv][lane * 8 + i] = ab[i];
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 256) {
        const int c = blockIdx.x * 512 + i;
        if (c < cols) atomicAdd(dbias + c, lds[0][i] + lds[1][i] + lds[2][i] + lds[3][i]);
    }
}

// ------------------------------------------------------------------ mask fill (+ backward)
template <typename T>
__global__ void mask_fill_rows_kernel(T* __restrict__ x, const uint8_t* __restrict__ mask, const float* __restrict__ emb, int rows, int cols) {
    const int nch = cols >> 3;
    const long long total = (long long)rows * nch;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / nch), ch = (int)(i % nch);
        if (mask[row]) {
            float e[8];
            load8(emb + ch * 8, e);
            store8(x + (long long)row * cols + ch * 8, e);
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void mask_fill_rows_bwd_kernel(T* __restrict__ dy, const uint8_t* __restrict__ mask, float* __restrict__ demb, int rows, int cols) {
    __shared__ __attribute__((aligned(16))) float lds[4 * MAXC * 512];
    const int lane = threadIdx.x & 63, nch = cols >> 3;
    float ab[MAXC][8];
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) ab[c][i] = 0.f;
    const float zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        if (!mask[row]) continue;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                const long long off = (long long)row * cols + ch * 8;
                float g[8];
                load8(dy + off, g);
#pragma unroll
                for (int i = 0; i < 8; ++i) ab[c][i] += g[i];
                store8(dy + off, zero);
            }
        }
    }
    flush_colsum<4, MAXC>(ab, lds, demb, cols);
}

template <typename T>
__global__ void mask_zero_cols_kernel(T* __restrict__ x, const uint8_t* __restrict__ cmask, int B, int Tn, int cols) {
    const long long total = (long long)B * Tn * cols;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cols);
        const int b = (int)(i / ((long long)Tn * cols));
        if (cmask[b * cols + c]) x[i] = from_f<T>(0.f);
    }
}
template <typename T>
__global__ void zero_rows_kernel(T* __restrict__ x, const uint8_t* __restrict__ keep, int rows, int cols) {
    const int nch = cols >> 3;
    const long long total = (long long)rows * nch;
    const float zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / nch), ch = (int)(i % nch);
        if (!keep[row]) store8(x + (long long)row * cols + ch * 8, zero);
    }
}

// ------------------------------------------------------------------ row gather / scatter-add
template <typename T, bool SCATTER>
__global__ void gather_rows_kernel(const T* __restrict__ src, const int32_t* __restrict__ idx, T* __restrict__ dst, int n, int cols) {
    const int nch = cols >> 3;
    const long long total = (long long)n * nch;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / nch), ch = (int)(i % nch);
        float v[8];
        if (!SCATTER) {
            load8(src + (long long)idx[r] * cols + ch * 8, v);
            store8(dst + (long long)r * cols + ch * 8, v);
        } else {
            float d[8];
            T* dp = dst + (long long)idx[r] * cols + ch * 8;
            load8(src + (long long)r * cols + ch * 8, v);
            load8(dp, d);
#pragma unroll
            for (int j = 0; j < 8; ++j) d[j] += v[j];
            store8(dp, d);
        }
    }
}

template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, long long n8) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        float u[8], v[8];
        load8(a + i * 8, u);
        load8(b + i * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) u[j] += v[j];
        store8(y + i * 8, u);
    }
}

template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, long long n8, Drop dr) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        float v[8];
        load8(x + i * 8, v);
        const Keep8 kp = a8_keep8(dr.seed, dr.stream, (uint32_t)i, dr.thr);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = a8_kept(kp, j) ? v[j] * dr.scale : 0.f;
        store8(y + i * 8, v);
    }
}

template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ s, D* __restrict__ d, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        d[i] = (D)(float)s[i];
}

// ------------------------------------------------------------------ col2im * gelu'(z) for the strided conv dgrad
template <typename T>
__global__ void col2im_kernel(const T* __restrict__ dcol, const T* __restrict__ z, T* __restrict__ dx,
                              int B, int Lin, int Lout, int Cc, int k, int stride) {
    const int nch = Cc >> 3;
    const long long total = (long long)B * Lin * nch;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % nch);
        const long long bp = i / nch;
        const int p = (int)(bp % Lin), b = (int)(bp / Lin);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int kk = 0; kk < k; ++kk) {
            const int q = p - kk;
            if (q < 0 || q % stride) continue;
            const int tt = q / stride;
            if (tt >= Lout) continue;
            float v[8];
            load8(dcol + ((long long)(b * Lout + tt) * k + kk) * Cc + ch * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += v[j];
        }
        const long long off = bp * Cc + ch * 8;
        if (z) {
            float zz[8];
            load8(z + off, zz);
            mul_gelu_grad_n(acc, zz);
        }
        store8(dx + off, acc);
    }
}

// ------------------------------------------------------------------ log-softmax over the label axis, one wave per row
// (any number of labels: a lane walks columns lane, lane + 64, ...; rows of x / dx may be padded to ldx elements)
template <typename T>
__global__ void log_softmax_fwd_kernel(const T* __restrict__ x, float* __restrict__ y, int rows, int cols, int ldx) {
    const int lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        const T* xr = x + (long long)row * ldx;
        float m = -INFINITY;
        for (int c = lane; c < cols; c += 64) m = fmaxf(m, to_f(xr[c]));
        m = wave_max(m);
        float s = 0.f;
        for (int c = lane; c < cols; c += 64) s += __expf(to_f(xr[c]) - m);
        const float lse = m + __logf(wave_sum(s));
        for (int c = lane; c < cols; c += 64) y[(long long)row * cols + c] = to_f(xr[c]) - lse;
    }
}
template <typename T>
__global__ void log_softmax_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, T* __restrict__ dx, int rows, int cols, int ldx) {
    const int lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        const long long o = (long long)row * cols;
        float s = 0.f;
        for (int c = lane; c < cols; c += 64) s += dy[o + c];
        s = wave_sum(s);
        for (int c = lane; c < cols; c += 64) dx[(long long)row * ldx + c] = from_f<T>(dy[o + c] - __expf(y[o + c]) * s);
        for (int c = cols + lane; c < ldx; c += 64) dx[(long long)row * ldx + c] = from_f<T>(0.f);      // padding columns carry no gradient
    }
}

// ------------------------------------------------------------------ grad-norm + AdamW
// Sum of squares in a FIXED order: every block writes its partial sum, a one-block kernel adds them in index order.  (One
// atomicAdd per block gave each data-parallel rank its own summation order: clip coefficients that differ in the last bit and
// replicas that drift apart by an ulp per step, although their reduced gradients are bit-identical.)
constexpr int SUMSQ_MAX_BLOCKS = 4096;
__device__ float g_sumsq_partial[SUMSQ_MAX_BLOCKS];       // per device (code object); one a8_sumsq in flight per device at a time

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long long n, float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    const long long n4 = n >> 2;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += gridDim.x * 256ll) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = x[n4 * 4 + threadIdx.x]; s += v * v; }
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partial, int nb, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) out[0] += s;
}

// NT: bit 0 = nontemporal loads, bit 1 = nontemporal stores (a8_set_tuning key 21; every byte is touched once per step)
template <int NT>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             long long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2s,
                             const float* __restrict__ gnorm_sq, float max_norm, float grad_scale, const long long* __restrict__ grad_denom,
                             int zero_grad) {
    if (grad_denom) grad_scale /= (float)grad_denom[0];       // scale_grads(world / sum of batch sizes), audio8/train.py:320-323
    float clip = grad_scale;
    if (gnorm_sq && max_norm > 0.f) {
        const float nrm = sqrtf(gnorm_sq[0]) * grad_scale;
        clip *= fminf(1.0f, max_norm / (nrm + 1e-6f));     // torch.nn.utils.clip_grad_norm_
    }
    const float step = lr / bc1, decay = 1.f - lr * wd;
    auto upd = [&](float& pi, float& gi, float& mi, float& vi) {
        const float gc = gi * clip;
        mi = b1 * mi + (1.f - b1) * gc;
        vi = b2 * vi + (1.f - b2) * gc * gc;
        pi = pi * decay - step * mi / (sqrtf(vi) / bc2s + eps);
        if (zero_grad) gi = 0.f;                           // the next step accumulates into a clean buffer: no separate fill pass
    };
    // 16 bytes per lane (the four flat buffers are 16-byte aligned and every range handed in starts on a multiple of 4 elements)
    const long long n4 = n >> 2;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        f32x4 P, G, M, V;
        if constexpr (NT & 1) {
            P = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(p) + i); G = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(g) + i);
            M = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(m) + i); V = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(v) + i);
        } else {
            P = reinterpret_cast<f32x4*>(p)[i]; G = reinterpret_cast<f32x4*>(g)[i]; M = reinterpret_cast<f32x4*>(m)[i]; V = reinterpret_cast<f32x4*>(v)[i];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float pe = P[e], ge = G[e], me = M[e], ve = V[e];
            upd(pe, ge, me, ve);
            P[e] = pe; G[e] = ge; M[e] = me; V[e] = ve;
        }
        if constexpr (NT & 2) {
            __builtin_nontemporal_store(P, reinterpret_cast<f32x4*>(p) + i);
            __builtin_nontemporal_store(M, reinterpret_cast<f32x4*>(m) + i);
            __builtin_nontemporal_store(V, reinterpret_cast<f32x4*>(v) + i);
            if (zero_grad) __builtin_nontemporal_store(G, reinterpret_cast<f32x4*>(g) + i);
        } else {
            reinterpret_cast<f32x4*>(p)[i] = P;
            reinterpret_cast<f32x4*>(m)[i] = M;
            reinterpret_cast<f32x4*>(v)[i] = V;
            if (zero_grad) reinterpret_cast<f32x4*>(g)[i] = G;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long long i = n4 * 4 + threadIdx.x;
        upd(p[i], g[i], m[i], v[i]);
    }
}

int grid_for(long long work, int per_block) {
    long long b = (work + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > 4096) b = 4096;
    return (int)b;
}

}  // namespace

#define DISPATCH_T(dtype, ...)                                             \
    do {                                                                   \
        if ((dtype) == A8_BF16) { typedef bf16 T; __VA_ARGS__; }           \
        else if ((dtype) == A8_F32) { typedef float T; __VA_ARGS__; }      \
        else return a8_set_error(A8_ERR_ARG, "bad dtype %d", (dtype));     \
    } while (0)

// rng_elem0: element offset of row 0 in the LOGICAL activation the dropout counters belong to (a multiple of 8): a call on rows
// [r0, r0 + rows) of a larger batch passes r0 * cols and draws exactly the mask the whole-batch call (and its backward) draws there.
int a8i_layernorm_fwd(int dtype, const void* x, const void* res, const float* gamma, const float* beta, void* y,
                      float* mean, float* rstd, int rows, int cols, float eps, float p, uint32_t seed, uint32_t rng_stream, int64_t rng_elem0,
                      void* stream) {
    A8_REQUIRE(cols % 8 == 0 && cols <= MAXC * 512, "a8_layernorm_fwd: cols=%d must be a multiple of 8 and <= %d", cols, MAXC * 512);
    A8_REQUIRE(p >= 0.f && p < 1.f, "a8_layernorm_fwd: bad dropout p");
    A8_REQUIRE(rng_elem0 % 8 == 0, "a8_layernorm_fwd: the counter offset must be a multiple of 8");
    if (rows == 0) return A8_OK;
    const Drop dr = make_drop(p, seed, rng_stream);
#define LN_FWD(MC) DISPATCH_T(dtype, hipLaunchKernelGGL((ln_fwd_kernel<T, MC>), dim3(grid_for(rows, 4)), dim3(256), 0, (hipStream_t)stream, \
                                         (const T*)x, (const T*)res, gamma, beta, (T*)y, mean, rstd, rows, cols, eps, dr, (long long)rng_elem0))
    if (cols <= 512) LN_FWD(1); else if (cols <= 1024) LN_FWD(2); else LN_FWD(MAXC);
#undef LN_FWD
    return a8_check_launch("a8_layernorm_fwd");
}
extern "C" int a8_layernorm_fwd(int dtype, const void* x, const void* res, const float* gamma, const float* beta, void* y,
                                float* mean, float* rstd, int rows, int cols, float eps, float p, uint32_t seed, uint32_t rng_stream, void* stream) {
    return a8i_layernorm_fwd(dtype, x, res, gamma, beta, y, mean, rstd, rows, cols, eps, p, seed, rng_stream, 0, stream);
}

// geometry of a launch: waves per block and rows per wave (a8_set_tuning key 1 = NW*100 + R for the atomic column sums: fewer, fatter
// blocks mean fewer same-address atomics; key 17 for the partial-row form, where nothing is contended and more, smaller blocks
// hide the row latency: 8 x 1 / 4 x 2 measured best, -0.06 ms per step against 8 x 2 with atomics)
static void ln_bwd_geometry(bool partials, int& nw, int& rpw) {
    const int t = partials ? g_a8_tuning[17] : g_a8_tuning[1];
    nw = t / 100; rpw = t % 100;
}
int a8i_layernorm_bwd_partial_rows(int rows) {
    int nw, rpw;
    ln_bwd_geometry(true, nw, rpw);
    return (nw == 4 || nw == 8 || nw == 16) && rpw > 0 ? cdiv(rows, nw * rpw) : 0;
}
int a8i_layernorm_bwd(int dtype, const void* dy, const void* dy2, const void* x, const void* res, const float* gamma,
                      const float* mean, const float* rstd, void* dx, void* dres, float* dgamma, float* dbeta, float* dbias,
                      int rows, int cols, float p, uint32_t seed, uint32_t rng_stream, int64_t rng_elem0, void* stream, float* partials) {
    A8_REQUIRE(rng_elem0 % 8 == 0, "a8_layernorm_bwd: the counter offset must be a multiple of 8");
    A8_REQUIRE(cols % 8 == 0 && cols <= MAXC * 512, "a8_layernorm_bwd: cols=%d must be a multiple of 8 and <= %d", cols, MAXC * 512);
    A8_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "a8_layernorm_bwd: dgamma and dbeta go together");
    if (rows == 0) return A8_OK;
    const Drop dr = make_drop(p, seed, rng_stream);
    // rows per wave R and waves per block NW (a8_set_tuning key 1 = NW*100 + R): fewer, fatter blocks mean fewer
    // same-address atomics on the column sums
    int nw, rpw;
    ln_bwd_geometry(partials != nullptr, nw, rpw);
    A8_REQUIRE(rpw > 0, "a8_layernorm_bwd: bad tuning");
    const int grid = cdiv(rows, nw * rpw);
#define LN_BWD2(NW, MC) DISPATCH_T(dtype, hipLaunchKernelGGL((ln_bwd_kernel<T, NW, MC>), dim3(grid), dim3(NW * 64), 0, (hipStream_t)stream, \
                                         (const T*)dy, (const T*)dy2, (const T*)x, (const T*)res, gamma, mean, rstd, (T*)dx, (T*)dres, \
                                         dgamma, dbeta, dbias, partials, rows, cols, dr, (long long)rng_elem0))
#define LN_BWD(NW) do { if (cols <= 1024) LN_BWD2(NW, 2); else LN_BWD2(NW, MAXC); } while (0)
    if (nw == 4) LN_BWD(4); else if (nw == 8) LN_BWD(8); else if (nw == 16) LN_BWD(16);
    else return a8_set_error(A8_ERR_ARG, "a8_layernorm_bwd: bad tuning %d", g_a8_tuning[1]);
#undef LN_BWD
#undef LN_BWD2
    return a8_check_launch("a8_layernorm_bwd");
}

extern "C" int a8_layernorm_bwd(int dtype, const void* dy, const void* dy2, const void* x, const void* res, const float* gamma,
                                const float* mean, const float* rstd, void* dx, void* dres, float* dgamma, float* dbeta, float* dbias,
                                int rows, int cols, float p, uint32_t seed, uint32_t rng_stream, void* stream) {
    return a8i_layernorm_bwd(dtype, dy, dy2, x, res, gamma, mean, rstd, dx, dres, dgamma, dbeta, dbias, rows, cols, p, seed, rng_stream, 0, stream, nullptr);
}

extern "C" int a8_layernorm_bwd_partial_rows(int rows) { return a8i_layernorm_bwd_partial_rows(rows); }
extern "C" int a8_colsum_partials_reduce(const float* ws, int nrows, int cols, float* out0, float* out1, float* out2, void* stream) {
    A8_REQUIRE(ws != nullptr && nrows >= 0 && cols > 0, "a8_colsum_partials_reduce: bad arguments");
    if (nrows == 0) return A8_OK;
    int gy = nrows < 64 ? nrows : 64;
    hipLaunchKernelGGL(colsum_partials_reduce_kernel, dim3(cdiv(3 * cols, 256), gy), dim3(256), 0, (hipStream_t)stream, ws, nrows, cols, out0, out1, out2);
    return a8_check_launch("a8_colsum_partials_reduce");
}

extern "C" int a8_gelu_bwd_bias(int dtype, const void* dy, const void* z, void* dz, float* dbias, int rows, int cols, void* stream) {
    A8_REQUIRE(cols % 8 == 0, "a8_gelu_bwd_bias: cols=%d must be a multiple of 8", cols);
    if (rows == 0) return A8_OK;
    const int gx = cdiv(cols, 512);
    int gy = cdiv(rows, 4 * 16);
    if (gy < 1) gy = 1;
    if (gy * gx > 2048) gy = 2048 / gx > 0 ? 2048 / gx : 1;
    DISPATCH_T(dtype, hipLaunchKernelGGL(gelu_bwd_bias_kernel<T>, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream,
                                         (const T*)dy, (const T*)z, (T*)dz, dbias, rows, cols));
    return a8_check_launch("a8_gelu_bwd_bias");
}

extern "C" int a8_mask_fill_rows(int dtype, void* x, const uint8_t* mask, const float* emb, int rows, int cols, void* stream) {
    A8_REQUIRE(cols % 8 == 0, "a8_mask_fill_rows: cols must be a multiple of 8");
    if (rows == 0) return A8_OK;
    DISPATCH_T(dtype, hipLaunchKernelGGL(mask_fill_rows_kernel<T>, dim3(grid_for((long long)rows * cols / 8, 256)), dim3(256), 0,
                                         (hipStream_t)stream, (T*)x, mask, emb, rows, cols));
    return a8_check_launch("a8_mask_fill_rows");
}
extern "C" int a8_mask_fill_rows_bwd(int dtype, void* dy, const uint8_t* mask, float* demb, int rows, int cols, void* stream) {
    A8_REQUIRE(cols % 8 == 0 && cols <= MAXC * 512, "a8_mask_fill_rows_bwd: cols=%d unsupported", cols);
    if (rows == 0) return A8_OK;
    DISPATCH_T(dtype, hipLaunchKernelGGL(mask_fill_rows_bwd_kernel<T>, dim3(grid_for(rows, 32)), dim3(256), 0, (hipStream_t)stream,
                                         (T*)dy, mask, demb, rows, cols));
    return a8_check_launch("a8_mask_fill_rows_bwd");
}
extern "C" int a8_mask_zero_cols(int dtype, void* x, const uint8_t* cmask, int B, int T_, int cols, void* stream) {
    if ((long long)B * T_ * cols == 0) return A8_OK;
    DISPATCH_T(dtype, hipLaunchKernelGGL(mask_zero_cols_kernel<T>, dim3(grid_for((long long)B * T_ * cols, 256)), dim3(256), 0,
                                         (hipStream_t)stream, (T*)x, cmask, B, T_, cols));
    return a8_check_launch("a8_mask_zero_cols");
}
extern "C" int a8_zero_rows(int dtype, void* x, const uint8_t* keep, int rows, int cols, void* stream) {
    A8_REQUIRE(cols % 8 == 0, "a8_zero_rows: cols must be a multiple of 8");
    if (rows == 0) return A8_OK;
    DISPATCH_T(dtype, hipLaunchKernelGGL(zero_rows_kernel<T>, dim3(grid_for((long long)rows * cols / 8, 256)), dim3(256), 0,
                                         (hipStream_t)stream, (T*)x, keep, rows, cols));
    return a8_check_launch("a8_zero_rows");
}
extern "C" int a8_gather_rows(int dtype, const void* src, const int32_t* idx, void* dst, int n, int cols, void* stream) {
    A8_REQUIRE(cols % 8 == 0, "a8_gather_rows: cols must be a multiple of 8");
    if (n == 0) return A8_OK;
    DISPATCH_T(dtype, hipLaunchKernelGGL((gather_rows_kernel<T, false>), dim3(grid_for((long long)n * cols / 8, 256)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)src, idx, (T*)dst, n, cols));
    return a8_check_launch("a8_gather_rows");
}
extern "C" int a8_scatter_add_rows(int dtype, const void* src, const int32_t* idx, void* dst, int n, int cols, void* stream) {
    A8_REQUIRE(cols % 8 == 0, "a8_scatter_add_rows: cols must be a multiple of 8");
    if (n == 0) return A8_OK;
    DISPATCH_T(dtype, hipLaunchKernelGGL((gather_rows_kernel<T, true>), dim3(grid_for((long long)n * cols / 8, 256)), dim3(256), 0,
                                         (hipStream_t)stream, (const T*)src, idx, (T*)dst, n, cols));
    return a8_check_launch("a8_scatter_add_rows");
}
extern "C" int a8_add(int dtype, const void* a, const void* b, void* y, int64_t n, void* stream) {
    A8_REQUIRE(n % 8 == 0, "a8_add: n must be a multiple of 8");
    if (n == 0) return A8_OK;
    DISPATCH_T(dtype, hipLaunchKernelGGL(add_kernel<T>, dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                                         (const T*)a, (const T*)b, (T*)y, (long long)(n / 8)));
    return a8_check_launch("a8_add");
}
extern "C" int a8_dropout(int dtype, const void* x, void* y, int64_t n, float p, uint32_t seed, uint32_t rng_stream, void* stream) {
    A8_REQUIRE(n % 8 == 0, "a8_dropout: n must be a multiple of 8");
    A8_REQUIRE(p >= 0.f && p < 1.f && n < (1ll << 32), "a8_dropout: bad p or n");
    if (n == 0) return A8_OK;
    const Drop dr = make_drop(p, seed, rng_stream);
    DISPATCH_T(dtype, hipLaunchKernelGGL(dropout_kernel<T>, dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                                         (const T*)x, (T*)y, (long long)(n / 8), dr));
    return a8_check_launch("a8_dropout");
}
extern "C" int a8_cast(int src_dtype, int dst_dtype, const void* src, void* dst, int64_t n, void* stream) {
    if (n == 0) return A8_OK;
    const dim3 g(grid_for(n, 256)), b(256);
    hipStream_t s = (hipStream_t)stream;
    if (src_dtype == A8_F32 && dst_dtype == A8_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16>), g, b, 0, s, (const float*)src, (bf16*)dst, (long long)n);
    else if (src_dtype == A8_BF16 && dst_dtype == A8_F32) hipLaunchKernelGGL((cast_kernel<bf16, float>), g, b, 0, s, (const bf16*)src, (float*)dst, (long long)n);
    else if (src_dtype == A8_F32 && dst_dtype == A8_F32) hipLaunchKernelGGL((cast_kernel<float, float>), g, b, 0, s, (const float*)src, (float*)dst, (long long)n);
    else if (src_dtype == A8_BF16 && dst_dtype == A8_BF16) hipLaunchKernelGGL((cast_kernel<bf16, bf16>), g, b, 0, s, (const bf16*)src, (bf16*)dst, (long long)n);
    else return a8_set_error(A8_ERR_ARG, "a8_cast: bad dtypes %d -> %d", src_dtype, dst_dtype);
    return a8_check_launch("a8_cast");
}
extern "C" int a8_col2im_gelu_grad(int dtype, const void* dcol, const void* z, void* dx, int B, int Lin, int Lout, int C_, int k, int stride, void* stream) {
    A8_REQUIRE(C_ % 8 == 0 && k >= 1 && stride >= 1, "a8_col2im_gelu_grad: bad shape");
    A8_REQUIRE((Lout - 1) * stride + k <= Lin, "a8_col2im_gelu_grad: Lout=%d does not fit Lin=%d", Lout, Lin);
    if ((long long)B * Lin == 0) return A8_OK;
    DISPATCH_T(dtype, hipLaunchKernelGGL(col2im_kernel<T>, dim3(grid_for((long long)B * Lin * C_ / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                                         (const T*)dcol, (const T*)z, (T*)dx, B, Lin, Lout, C_, k, stride));
    return a8_check_launch("a8_col2im_gelu_grad");
}
extern "C" int a8_log_softmax_fwd(const void* x, int dtype, float* y, int rows, int cols, int ldx, void* stream) {
    A8_REQUIRE(cols >= 1 && ldx >= cols, "a8_log_softmax_fwd: cols=%d must be >= 1 and ldx=%d >= cols", cols, ldx);
    if (rows == 0) return A8_OK;
    DISPATCH_T(dtype, hipLaunchKernelGGL(log_softmax_fwd_kernel<T>, dim3(grid_for(rows, 4)), dim3(256), 0, (hipStream_t)stream, (const T*)x, y, rows, cols, ldx));
    return a8_check_launch("a8_log_softmax_fwd");
}
extern "C" int a8_log_softmax_bwd(const float* y, const float* dy, void* dx, int dtype, int rows, int cols, int ldx, void* stream) {
    A8_REQUIRE(cols >= 1 && ldx >= cols, "a8_log_softmax_bwd: cols=%d must be >= 1 and ldx=%d >= cols", cols, ldx);
    if (rows == 0) return A8_OK;
    DISPATCH_T(dtype, hipLaunchKernelGGL(log_softmax_bwd_kernel<T>, dim3(grid_for(rows, 4)), dim3(256), 0, (hipStream_t)stream, y, dy, (T*)dx, rows, cols, ldx));
    return a8_check_launch("a8_log_softmax_bwd");
}
extern "C" int a8_sumsq(const float* x, int64_t n, float* out, void* stream) {
    if (n == 0) return A8_OK;
    A8_REQUIRE(((uintptr_t)x % 16) == 0, "a8_sumsq: x must be 16-byte aligned");
    const int nb = grid_for(n / 4 + 1, 1024);             // <= SUMSQ_MAX_BLOCKS
    float* partial = nullptr;
    if (hipGetSymbolAddress((void**)&partial, HIP_SYMBOL(g_sumsq_partial)) != hipSuccess) return a8_set_error(A8_ERR_LAUNCH, "a8_sumsq: no partial buffer");
    hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, (long long)n, partial);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)partial, nb, out);
    return a8_check_launch("a8_sumsq");
}
extern "C" int a8_adamw(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int step, const float* gnorm_sq, float max_norm, float grad_scale, const int64_t* grad_denom,
                        int zero_grad, void* stream) {
    A8_REQUIRE(step >= 1, "a8_adamw: step counts from 1");
    if (n == 0) return A8_OK;
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
    A8_REQUIRE(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0,
               "a8_adamw: buffers must be 16-byte aligned");
    // Launch geometry (profiles/r05_adamw.txt): a 4096-block grid-stride loop left HBM latency exposed (5.6 TB/s on a box whose copy
    // rate is 6.3); one 16-byte group per thread up to 65536 blocks (then grid-stride) and nontemporal loads + stores (every byte is
    // touched exactly once per step) reach 6.4 TB/s on the same box: 544 -> 476 us for the base model's 95 M parameters.
    // a8_set_tuning key 20 = grid in blocks (0: this default), key 21 = cache policy (0: default = 3; 1 loads only, 2 stores only, 3 both, 4 neither)
    const long long want = n / 4 / 256 + 1;
    const int grid = (int)std::min<long long>(g_a8_tuning[20] > 0 ? g_a8_tuning[20] : 65536, want);
    const int pol = g_a8_tuning[21] == 0 ? 3 : (g_a8_tuning[21] & 3);
#define A8_ADAMW(NT_) hipLaunchKernelGGL(adamw_kernel<NT_>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n, lr, beta1, beta2, eps, \
                                        weight_decay, bc1, bc2s, gnorm_sq, max_norm, grad_scale, (const long long*)grad_denom, zero_grad)
    switch (pol) {
        case 1: A8_ADAMW(1); break;
        case 2: A8_ADAMW(2); break;
        case 3: A8_ADAMW(3); break;
        default: A8_ADAMW(0); break;
    }
#undef A8_ADAMW
    return a8_check_launch("a8_adamw");
}
