// a8hip: Gumbel-softmax product quantiser (GumbelVectorQuantizer.forward, audio8/wav2vec2.py:547-576) and the
// negatives-gather + cosine + cross-entropy loss (Sampler.negatives :955-976, Wav2Vec2Loss :371-392).
// Wavefront-shuffle kernels: one wave owns one row (one (frame, group) of logits / one masked frame),
// all reductions are xor-shuffles, nothing of size [rows, G*V, D] or [N, B, Tm, C] is ever materialised
// (the reference's broadcast codebook product is 241 M elements, its negatives tensor 75 M).
#include "a8_common.h"
#include "a8hip_internal.h"

namespace {

constexpr int VMAX_PER_LANE = 8;   // V <= 512

// ------------------------------------------------------------------ VQ: argmax + mean softmax
// One row per wave, 16 rows per block: the block's 16 softmax rows are summed through LDS and written as ONE partial
// row (no atomics: R*V same-address adds, LDS or global, were 25 of this kernel's 30 us); vq_avg_reduce_kernel sums the
// partial rows into avg_probs.
constexpr int VQ_ROWS = 16;
// gumbel_gen != nullptr (and gumbel == nullptr): the Gumbel(0,1) noise of F.gumbel_softmax (-log E, E ~ Exp(1): torch's
// `-empty.exponential_().log()`) is drawn here from a counter hash of (seed, element index) and written to gumbel_gen for backward.
__global__ __launch_bounds__(64 * VQ_ROWS) void vq_select_kernel(const float* __restrict__ logits, const float* __restrict__ gumbel,
                                                                 float* __restrict__ gumbel_gen, uint32_t seed,
                                                                 int32_t* __restrict__ codes, float* __restrict__ partial, int R, int V) {
    __shared__ float s_p[VQ_ROWS * 64 * VMAX_PER_LANE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int row = blockIdx.x * VQ_ROWS + wv;
    float x[VMAX_PER_LANE];
    float inv = 0.f;
    if (row < R) {
        float mx = -INFINITY, best = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < VMAX_PER_LANE; ++j) {
            const int v = lane + 64 * j;
            x[j] = v < V ? logits[(long long)row * V + v] : -INFINITY;
            mx = fmaxf(mx, x[j]);
            if (v < V) {
                float y = x[j];
                if (gumbel) y += gumbel[(long long)row * V + v];
                else if (gumbel_gen) {
                    const uint32_t h = a8_hash(seed, 0x6756u, (uint32_t)((long long)row * V + v));
                    // uniform in (0, 1) from 23 bits: (k + 0.5) / 2^23 is exact in f32 for every k < 2^23 -- with 24 bits the top value
                    // rounds to 1.0, whose noise is -log(-log 1) = +inf (an inf - inf = NaN in the straight-through softmax of backward)
                    const float u = ((float)(h >> 9) + 0.5f) * 1.1920928955078125e-07f;
                    const float g = -__logf(-__logf(u));
                    gumbel_gen[(long long)row * V + v] = g;
                    y += g;
                }
                if (y > best) { best = y; bi = v; }
            }
        }
        mx = wave_max(mx);
        // argmax with first-index tie-break (torch.max semantics)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < VMAX_PER_LANE; ++j) { x[j] = (lane + 64 * j < V) ? __expf(x[j] - mx) : 0.f; s += x[j]; }
        s = wave_sum(s);
        inv = 1.0f / ((float)R * s);
        if (lane == 0) codes[row] = (unsigned)bi < (unsigned)V ? bi : 0;      // (all-NaN logits compare false everywhere: never hand an out-of-range
                                                                               //  index to the codebook gather)
    }
#pragma unroll
    for (int j = 0; j < VMAX_PER_LANE; ++j)
        if (lane + 64 * j < V) s_p[wv * V + lane + 64 * j] = row < R ? x[j] * inv : 0.f;
    __syncthreads();
    for (int v = threadIdx.x; v < V; v += 64 * VQ_ROWS) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < VQ_ROWS; ++q) t += s_p[q * V + v];
        partial[(long long)blockIdx.x * V + v] = t;
    }
}

// avg[v] = sum_b partial[b][v]; block c owns columns [64 c, 64 c + 64), 16 row groups
__global__ __launch_bounds__(1024) void vq_avg_reduce_kernel(const float* __restrict__ partial, float* __restrict__ avg, int nb, int V) {
    __shared__ float red[1024];
    const int r = threadIdx.x >> 6, v = blockIdx.x * 64 + (threadIdx.x & 63);
    float t = 0.f;
    if (v < V)
        for (int b = r; b < nb; b += 16) t += partial[(long long)b * V + v];
    red[threadIdx.x] = t;
    __syncthreads();
    if (r == 0 && v < V) {
#pragma unroll
        for (int q = 1; q < 16; ++q) t += red[q * 64 + (threadIdx.x & 63)];
        avg[v] = t;
    }
}

template <typename T>
__global__ void vq_gather_kernel(const float* __restrict__ vars, const int32_t* __restrict__ codes, T* __restrict__ q, int N, int G, int V, int D) {
    const long long total = (long long)N * G * D;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int dd = (int)(i % D);
        const long long ng = i / D;
        const int g = (int)(ng % G);
        q[i] = from_f<T>(vars[((long long)g * V + codes[ng]) * D + dd]);
    }
}

// prob_ppl = exp(-sum p log(p + 1e-7)),  dppl/dp_v = -ppl * (log(p_v + 1e-7) + p_v / (p_v + 1e-7))
__global__ __launch_bounds__(256) void vq_ppl_kernel(const float* __restrict__ avg, float* __restrict__ ppl, float* __restrict__ dppl, int V) {
    __shared__ float red[4];
    float h = 0.f;
    for (int i = threadIdx.x; i < V; i += 256) h += avg[i] * logf(avg[i] + 1e-7f);
    h = block_sum<4>(h, red);
    const float pp = expf(-h);
    if (threadIdx.x == 0) ppl[0] = pp;
    for (int i = threadIdx.x; i < V; i += 256) dppl[i] = -pp * (logf(avg[i] + 1e-7f) + avg[i] / (avg[i] + 1e-7f));
}

// backward through the straight-through soft path and the diversity path; one wave per (frame, group) row
template <typename T>
__global__ __launch_bounds__(256) void vq_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ gumbel, float tau,
                                                     const int32_t* __restrict__ codes, const float* __restrict__ vars,
                                                     const T* __restrict__ dq, const float* __restrict__ eproj, const float* __restrict__ dppl, const float* __restrict__ gppl,
                                                     float* __restrict__ dlogits, float* __restrict__ dvars,
                                                     int R, int G, int V, int D, int training) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float gp = gppl ? gppl[0] / (float)R : 0.f;
    const float itau = 1.0f / tau;
    for (int row = blockIdx.x * 4 + wv; row < R; row += gridDim.x * 4) {
        const int g = row % G;
        const long long n = row / G;
        // dq_g slice of this row: D <= 128 -> 2 per lane
        float dqv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) dqv[j] = (lane + 64 * j < D) ? to_f(dq[(n * G + g) * D + lane + 64 * j]) : 0.f;
        // codebook gradient: only the chosen entry (value of the straight-through one-hot is exactly the hard one-hot)
        const int code = codes[row];
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (lane + 64 * j < D) atomicAdd(dvars + ((long long)g * V + code) * D + lane + 64 * j, dqv[j]);
        float x[VMAX_PER_LANE], ys[VMAX_PER_LANE], e[VMAX_PER_LANE];
        float mx = -INFINITY, my = -INFINITY;
#pragma unroll
        for (int j = 0; j < VMAX_PER_LANE; ++j) {
            const int v = lane + 64 * j;
            x[j] = v < V ? logits[(long long)row * V + v] : -INFINITY;
            ys[j] = (v < V && training) ? (x[j] + gumbel[(long long)row * V + v]) * itau : -INFINITY;
            mx = fmaxf(mx, x[j]);
            my = fmaxf(my, ys[j]);
            e[j] = 0.f;
        }
        mx = wave_max(mx);
        float sx = 0.f, sy = 0.f;
#pragma unroll
        for (int j = 0; j < VMAX_PER_LANE; ++j) { x[j] = (lane + 64 * j < V) ? __expf(x[j] - mx) : 0.f; sx += x[j]; }
        sx = wave_sum(sx);
        float dpp = 0.f;      // sum_v p_v dppl_v
#pragma unroll
        for (int j = 0; j < VMAX_PER_LANE; ++j) {
            x[j] /= sx;
            if (lane + 64 * j < V) dpp += x[j] * dppl[lane + 64 * j];
        }
        dpp = wave_sum(dpp);
        float se = 0.f;
        if (training) {
            my = wave_max(my);
#pragma unroll
            for (int j = 0; j < VMAX_PER_LANE; ++j) { ys[j] = (lane + 64 * j < V) ? __expf(ys[j] - my) : 0.f; sy += ys[j]; }
            sy = wave_sum(sy);
            // e_v = <dq_g, vars[g, v, :]> was computed by a8_gemm (eproj [R, V])
#pragma unroll
            for (int j = 0; j < VMAX_PER_LANE; ++j)
                if (lane + 64 * j < V) e[j] = eproj[(long long)row * V + lane + 64 * j];
#pragma unroll
            for (int j = 0; j < VMAX_PER_LANE; ++j) { ys[j] /= sy; se += ys[j] * e[j]; }
            se = wave_sum(se);
        }
#pragma unroll
        for (int j = 0; j < VMAX_PER_LANE; ++j) {
            const int v = lane + 64 * j;
            if (v < V) {
                float gl = gp * x[j] * (dppl[v] - dpp);
                if (training) gl += itau * ys[j] * (e[j] - se);
                dlogits[(long long)row * V + v] = gl;
            }
        }
    }
}

// ------------------------------------------------------------------ contrastive: cosine logits + CE
// one wave per masked frame, C = 256: lane l owns channels {l, l+64, l+128, l+192}, so every load, and every
// f32 atomic of the backward scatter, is one 256-byte contiguous wave-instruction (the shape the memory-side
// atomic units run at full rate for; a 64-byte lane stride is ~17x slower).
constexpr int CPL = 4;      // channels per lane (C = 256)
template <typename T> A8_DEV void load_row(const T* p, int lane, float (&v)[CPL]) {
#pragma unroll
    for (int i = 0; i < CPL; ++i) v[i] = to_f(p[lane + 64 * i]);
}

// inverse row norms of both operands in one launch: inv[0..rows) for x0, inv[rows..2 rows) for x1
template <typename T>
__global__ __launch_bounds__(256) void rownorm_kernel(const T* __restrict__ x0, const T* __restrict__ x1, float* __restrict__ inv, int rows, int Cn) {
    const int lane = threadIdx.x & 63;
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < 2 * rows; r += gridDim.x * 4) {
        const T* x = r < rows ? x0 + (long long)r * Cn : x1 + (long long)(r - rows) * Cn;
        float s = 0.f;
        for (int c = lane; c < Cn; c += 64) { const float v = to_f(x[c]); s = fmaf(v, v, s); }
        s = wave_sum(s);
        if (lane == 0) inv[r] = 1.0f / fmaxf(sqrtf(s), 1e-8f);   // torch.cosine_similarity eps
    }
}

template <typename T>
__global__ __launch_bounds__(256) void contrastive_fwd_kernel(const T* __restrict__ ctx, const T* __restrict__ lat, const int32_t* __restrict__ idx,
                                                              const float* __restrict__ inv_ctx, const float* __restrict__ inv_lat,
                                                              float* __restrict__ logits, float* __restrict__ loss_sum, int rows, int N) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float lsum = 0.f;
    for (int row = blockIdx.x * 4 + wv; row < rows; row += gridDim.x * 4) {
        float c[CPL];
        load_row(ctx + (long long)row * 256, lane, c);
        const float ic = inv_ctx[row];
        float mine0 = -INFINITY, mine1 = -INFINITY;        // logits n = lane and n = lane + 64
        // candidate rows of this frame: lane n holds candidate n (0 = the frame's own latent) and n + 64, so the loop below reads its
        // row indices from registers and keeps FOUR candidates' loads in flight (same arithmetic per candidate: bit-identical logits;
        // one candidate at a time behind an index load was a chain of N + 1 dependent L2 round trips: 47 -> 2x us at cfg2)
        const int ci0 = lane == 0 ? row : (lane <= N ? idx[(long long)row * N + lane - 1] : row);
        const int ci1 = lane + 64 <= N ? idx[(long long)row * N + lane + 63] : row;
        auto cand = [&](int n) { return n < 64 ? __builtin_amdgcn_readlane(ci0, n) : __builtin_amdgcn_readlane(ci1, n - 64); };
        auto keep = [&](int n, float d) {
            if (n == lane) mine0 = d;
            if (n == lane + 64) mine1 = d;
        };
        int n = 0;
        for (; n + 4 <= N + 1; n += 4) {
            int tr[4];
            float tv[4][CPL], d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { tr[u] = cand(n + u); load_row(lat + (long long)tr[u] * 256, lane, tv[u]); }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                d[u] = 0.f;
#pragma unroll
                for (int i = 0; i < CPL; ++i) d[u] = fmaf(c[i], tv[u][i], d[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) keep(n + u, wave_sum(d[u]) * ic * inv_lat[tr[u]]);
        }
        for (; n <= N; ++n) {
            const int tr = cand(n);
            float tv[CPL];
            load_row(lat + (long long)tr * 256, lane, tv);
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < CPL; ++i) d = fmaf(c[i], tv[i], d);
            keep(n, wave_sum(d) * ic * inv_lat[tr]);
        }
        if (lane <= N) logits[(long long)row * (N + 1) + lane] = mine0;
        if (lane + 64 <= N) logits[(long long)row * (N + 1) + lane + 64] = mine1;
        const float mx = wave_max(fmaxf(mine0, mine1));
        float s = (lane <= N ? __expf(mine0 - mx) : 0.f) + (lane + 64 <= N ? __expf(mine1 - mx) : 0.f);
        s = wave_sum(s);
        const float pos = __shfl(mine0, 0, 64);
        if (lane == 0) lsum += mx + __logf(s) - pos;
    }
    lsum = block_sum<4>(lsum, red);
    if (threadIdx.x == 0) atomicAdd(loss_sum, lsum);
}

// Backward, two kernels.
//  rows kernel (one wave per masked frame): softmax weights dl[n]; dctx = sum_n dl*(t_n*ic*it_n - cos_n*c*ic^2); stores the
//  per-pair coefficients wA = dl*ic and wU = dl*cos.  The gradient of candidate row j is then
//      dlat[j] = it_j * sum_{(row,n)->j} wA*c_row  -  t_j * it_j^2 * sum_{(row,n)->j} wU
//  A frame's negatives come from its own utterance (rows_per_group consecutive rows), so the first sum is a small dense
//  product per utterance: scatter wA into W_b[frame][candidate] (3e5 scalar atomics), then G_b = W_b^T C_b as an exact-f32
//  grouped GEMM, then one combine pass -- instead of 7.6e7 global f32 atomics at 8 x 368 frames x 101 candidates.
//  (Accumulating the rows in LDS with ds_add_f32 was tried first: ~0.3 atomics per clock per CU, 3x slower than the
//  global scatter.)  Pairs that point outside the group (never produced by Sampler) are scattered directly.
template <typename T, bool SCATTER>
__global__ __launch_bounds__(256) void contrastive_bwd_kernel(const T* __restrict__ ctx, const T* __restrict__ lat, const int32_t* __restrict__ idx,
                                                              const float* __restrict__ inv_ctx, const float* __restrict__ inv_lat,
                                                              const float* __restrict__ logits, const float* __restrict__ gscale,
                                                              float* __restrict__ dctx, float* __restrict__ dlat,
                                                              float* __restrict__ wA, float* __restrict__ wU, int rows, int N) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float gs = gscale[0];
    for (int row = blockIdx.x * 4 + wv; row < rows; row += gridDim.x * 4) {
        const float l0 = lane <= N ? logits[(long long)row * (N + 1) + lane] : -INFINITY;
        const float l1 = lane + 64 <= N ? logits[(long long)row * (N + 1) + lane + 64] : -INFINITY;
        const float mx = wave_max(fmaxf(l0, l1));
        const float e0 = lane <= N ? __expf(l0 - mx) : 0.f, e1 = lane + 64 <= N ? __expf(l1 - mx) : 0.f;
        const float inv_s = 1.0f / wave_sum(e0 + e1);
        float c[CPL], dc[CPL];
        load_row(ctx + (long long)row * 256, lane, c);
#pragma unroll
        for (int i = 0; i < CPL; ++i) dc[i] = 0.f;
        const float ic = inv_ctx[row];
        if constexpr (!SCATTER) {
            const float d0 = gs * (e0 * inv_s - (lane == 0 ? 1.f : 0.f)), d1 = gs * e1 * inv_s;
            if (lane <= N) { wA[(long long)row * (N + 1) + lane] = d0 * ic; wU[(long long)row * (N + 1) + lane] = d0 * l0; }
            if (lane + 64 <= N) { wA[(long long)row * (N + 1) + lane + 64] = d1 * ic; wU[(long long)row * (N + 1) + lane + 64] = d1 * l1; }
        }
        // candidate rows from registers, four candidates' loads in flight (as in the forward; same accumulation order per channel)
        const int ci0 = lane == 0 ? row : (lane <= N ? idx[(long long)row * N + lane - 1] : row);
        const int ci1 = lane + 64 <= N ? idx[(long long)row * N + lane + 63] : row;
        auto cand = [&](int n) { return n < 64 ? __builtin_amdgcn_readlane(ci0, n) : __builtin_amdgcn_readlane(ci1, n - 64); };
        auto one = [&](int n, int tr, const float (&tv)[CPL]) {
            const float lg = n < 64 ? __shfl(l0, n, 64) : __shfl(l1, n - 64, 64);
            const float en = n < 64 ? __shfl(e0, n, 64) : __shfl(e1, n - 64, 64);
            const float dl = gs * (en * inv_s - (n == 0 ? 1.f : 0.f));
            const float it = inv_lat[tr];
            // d cos/dc = t*ic*it - cos*c*ic^2 ; d cos/dt = c*ic*it - cos*t*it^2
            const float a = dl * ic * it, bc = dl * lg * ic * ic, bt = dl * lg * it * it;
            float* dlp = dlat + (long long)tr * 256;
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                dc[i] += a * tv[i] - bc * c[i];
                if constexpr (SCATTER) atomicAdd(dlp + lane + 64 * i, a * c[i] - bt * tv[i]);
            }
        };
        int n = 0;
        for (; n + 4 <= N + 1; n += 4) {
            int tr[4];
            float tv[4][CPL];
#pragma unroll
            for (int u = 0; u < 4; ++u) { tr[u] = cand(n + u); load_row(lat + (long long)tr[u] * 256, lane, tv[u]); }
#pragma unroll
            for (int u = 0; u < 4; ++u) one(n + u, tr[u], tv[u]);
        }
        for (; n <= N; ++n) {
            const int tr = cand(n);
            float tv[CPL];
            load_row(lat + (long long)tr * 256, lane, tv);
            one(n, tr, tv);
        }
        float* dp = dctx + (long long)row * 256;
#pragma unroll
        for (int i = 0; i < CPL; ++i) dp[lane + 64 * i] = dc[i];
    }
}

// dense per-group coefficient matrix W[grp][frame][candidate row in group] += wA, U[row] += wU  (one thread per pair;
// ~3e5 scalar atomics instead of 7.6e7 row-element atomics); pairs outside the group are scattered directly.
template <typename T>
__global__ void contrastive_w_kernel(const T* __restrict__ ctx, const T* __restrict__ lat, const int32_t* __restrict__ idx,
                                     const float* __restrict__ inv_lat, const float* __restrict__ wA, const float* __restrict__ wU,
                                     float* __restrict__ W, float* __restrict__ U, float* __restrict__ dlat, int rows, int rpg, int ldw, int N) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= (long long)rows * (N + 1)) return;
    const int row = (int)(i / (N + 1)), n = (int)(i % (N + 1));
    const int tr = n == 0 ? row : idx[(long long)row * N + n - 1];
    const int base = (row / rpg) * rpg, jl = tr - base;
    const float a = wA[i], uu = wU[i];
    if (jl >= 0 && jl < rpg) {
        atomicAdd(W + (long long)row * ldw + jl, a);
        atomicAdd(U + tr, uu);
    } else {
        const float it = inv_lat[tr];
        for (int d = 0; d < 256; ++d)
            atomicAdd(dlat + (long long)tr * 256 + d, it * a * to_f(ctx[(long long)row * 256 + d]) - to_f(lat[(long long)tr * 256 + d]) * it * it * uu);
    }
}
// dlat[j] += it_j * G[j] - t_j * it_j^2 * U[j]   with G = W^T C from the GEMM
template <typename T>
__global__ void contrastive_combine_kernel(const T* __restrict__ lat, const float* __restrict__ inv_lat, const float* __restrict__ G,
                                           const float* __restrict__ U, float* __restrict__ dlat, long long n, int rpg, int rpg4) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long j = i >> 8;
    const float it = inv_lat[j];
    const long long gi = ((j / rpg) * rpg4 + j % rpg) * 256 + (i & 255);       // G keeps rpg4 (padded) rows per group
    dlat[i] += it * G[gi] - to_f(lat[i]) * it * it * U[j];
}

// loss = w_xe * (xe_sum / rows) + w_div * (n_vars - ppl) / n_vars, in the reference's operation order
// (audio8/wav2vec2.py:390-392); backward hands each branch its scalar cotangent.  One lane: these are scalars.
__global__ void loss_combine_kernel(const float* __restrict__ xe_sum, const float* __restrict__ ppl, float rows, float n_vars,
                                    float w_xe, float w_div, float* __restrict__ loss) {
    const float xe = xe_sum[0] / rows;
    const float diversity = w_div * (n_vars - ppl[0]) / n_vars;
    loss[0] = w_xe * xe + diversity;
}
__global__ void loss_combine_bwd_kernel(const float* __restrict__ dloss, float rows, float n_vars, float w_xe, float w_div,
                                        float* __restrict__ dxe_sum, float* __restrict__ dppl) {
    const float g = dloss[0];
    dxe_sum[0] = g * w_xe / rows;
    dppl[0] = -(g * w_div) / n_vars;
}

int gridw(int rows) { int b = (rows + 3) / 4; return b < 1 ? 1 : (b > 2048 ? 2048 : b); }

}  // namespace

extern "C" int a8_vq_select(const float* logits, const float* gumbel, float* gumbel_gen, uint32_t seed, float tau, int32_t* codes,
                            float* avg_probs, float* scratch, int R, int V, void* stream) {
    A8_REQUIRE(!(gumbel && gumbel_gen), "a8_vq_select: give the noise or ask for it, not both");
    A8_REQUIRE(V >= 1 && V <= 64 * VMAX_PER_LANE, "a8_vq_select: V=%d must be <= %d", V, 64 * VMAX_PER_LANE);
    A8_REQUIRE(tau > 0.f, "a8_vq_select: tau must be positive");
    A8_REQUIRE(R >= 0, "a8_vq_select: bad row count");
    const int nb = cdiv(R, VQ_ROWS);                    // (R = 0: avg_probs = 0)
    if (nb) hipLaunchKernelGGL(vq_select_kernel, dim3(nb), dim3(64 * VQ_ROWS), 0, (hipStream_t)stream, logits, gumbel, gumbel_gen, seed, codes, scratch, R, V);
    hipLaunchKernelGGL(vq_avg_reduce_kernel, dim3(cdiv(V, 64)), dim3(1024), 0, (hipStream_t)stream, (const float*)scratch, avg_probs, nb, V);
    return a8_check_launch("a8_vq_select");
}

extern "C" int a8_vq_gather(int dtype, const float* vars, const int32_t* codes, void* q, int N, int G, int V, int D, void* stream) {
    if ((long long)N * G * D == 0) return A8_OK;
    const long long total = (long long)N * G * D;
    const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    if (dtype == A8_BF16) hipLaunchKernelGGL(vq_gather_kernel<bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, vars, codes, (bf16*)q, N, G, V, D);
    else if (dtype == A8_F32) hipLaunchKernelGGL(vq_gather_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, vars, codes, (float*)q, N, G, V, D);
    else return a8_set_error(A8_ERR_ARG, "a8_vq_gather: bad dtype");
    return a8_check_launch("a8_vq_gather");
}

extern "C" int a8_vq_ppl(const float* avg_probs, float* ppl, float* dppl_davg, int V, void* stream) {
    hipLaunchKernelGGL(vq_ppl_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, avg_probs, ppl, dppl_davg, V);
    return a8_check_launch("a8_vq_ppl");
}

extern "C" int a8_vq_bwd(int dtype, const float* logits, const float* gumbel, float tau, const int32_t* codes, const float* vars,
                         const void* dq, const float* eproj, const float* dppl_davg, const float* gppl, float* dlogits, float* dvars,
                         int N, int G, int V, int D, int training, void* stream) {
    A8_REQUIRE(V >= 1 && V <= 64 * VMAX_PER_LANE && D <= 128, "a8_vq_bwd: V=%d D=%d unsupported", V, D);
    A8_REQUIRE(!training || (gumbel && eproj), "a8_vq_bwd: training needs the gumbel noise used in forward and eproj");
    const int R = N * G;
    if (R == 0) return A8_OK;
    if (dtype == A8_BF16) hipLaunchKernelGGL(vq_bwd_kernel<bf16>, dim3(gridw(R)), dim3(256), 0, (hipStream_t)stream, logits, gumbel, tau, codes, vars, (const bf16*)dq, eproj, dppl_davg, gppl, dlogits, dvars, R, G, V, D, training);
    else if (dtype == A8_F32) hipLaunchKernelGGL(vq_bwd_kernel<float>, dim3(gridw(R)), dim3(256), 0, (hipStream_t)stream, logits, gumbel, tau, codes, vars, (const float*)dq, eproj, dppl_davg, gppl, dlogits, dvars, R, G, V, D, training);
    else return a8_set_error(A8_ERR_ARG, "a8_vq_bwd: bad dtype");
    return a8_check_launch("a8_vq_bwd");
}

extern "C" int a8_contrastive_fwd(int dtype, const void* ctx, const void* lat, const int32_t* neg_idx, float* logits,
                                  float* loss_sum, float* inv_norms, int rows, int N, int Cn, void* stream) {
    A8_REQUIRE(Cn == 64 * CPL, "a8_contrastive_fwd: C must be %d (got %d)", 64 * CPL, Cn);
    A8_REQUIRE(N >= 1 && N + 1 <= 128, "a8_contrastive_fwd: N=%d must be in [1,127]", N);
    if (rows == 0) return A8_OK;
    float* ic = inv_norms; float* il = inv_norms + rows;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == A8_BF16) {
        hipLaunchKernelGGL(rownorm_kernel<bf16>, dim3(gridw(2 * rows)), dim3(256), 0, s, (const bf16*)ctx, (const bf16*)lat, ic, rows, Cn);
        hipLaunchKernelGGL(contrastive_fwd_kernel<bf16>, dim3(gridw(rows)), dim3(256), 0, s, (const bf16*)ctx, (const bf16*)lat, neg_idx, ic, il, logits, loss_sum, rows, N);
    } else if (dtype == A8_F32) {
        hipLaunchKernelGGL(rownorm_kernel<float>, dim3(gridw(2 * rows)), dim3(256), 0, s, (const float*)ctx, (const float*)lat, ic, rows, Cn);
        hipLaunchKernelGGL(contrastive_fwd_kernel<float>, dim3(gridw(rows)), dim3(256), 0, s, (const float*)ctx, (const float*)lat, neg_idx, ic, il, logits, loss_sum, rows, N);
    } else return a8_set_error(A8_ERR_ARG, "a8_contrastive_fwd: bad dtype");
    return a8_check_launch("a8_contrastive_fwd");
}

extern "C" int a8_contrastive_bwd(int dtype, const void* ctx, const void* lat, const int32_t* neg_idx, const float* logits,
                                  const float* gscale, const float* inv_norms, float* dctx, float* dlat, float* scratch, int rows, int N, int Cn,
                                  int rows_per_group, void* stream) {
    A8_REQUIRE(Cn == 64 * CPL, "a8_contrastive_bwd: C must be %d (got %d)", 64 * CPL, Cn);
    A8_REQUIRE(N >= 1 && N + 1 <= 128, "a8_contrastive_bwd: N=%d must be in [1,127]", N);
    A8_REQUIRE(dtype == A8_BF16 || dtype == A8_F32, "a8_contrastive_bwd: bad dtype");
    if (rows == 0) return A8_OK;
    const int rpg = rows_per_group;
    // (the bf16 copy of W reuses the Cf slab; below 8 rows per group the padded G slab [rows/rpg][rpg4][256] would outgrow the
    // 2*256 floats per row the scratch macro reserves -- such tiny groups take the scatter path)
    const bool grouped = rpg >= 8 && rows % rpg == 0 && (dtype != A8_BF16 || rpg <= 504);
    const int rpg4 = (rpg + 7) & ~7;                     // the TN GEMM wants M % 4 (f32) / % 8 (bf16) == 0: W rows are padded with zero columns
    // scratch: ic[rows] il[rows] wA[rows*(N+1)] wU[rows*(N+1)] | U[rows] W[rows*rpg] Cf[rows*256] G[rows*256]
    auto al4 = [](size_t n) { return (n + 3) & ~(size_t)3; };              // sub-buffers start 16-byte aligned (GEMM operands)
    A8_REQUIRE(((uintptr_t)scratch & 15) == 0, "a8_contrastive_bwd: scratch must be 16-byte aligned");
    // inverse row norms: the forward's (inv_norms, [2][rows]) or recomputed into scratch (then contiguous too)
    float* icw = scratch;
    const float* ic = inv_norms ? inv_norms : icw;
    const float* il = ic + rows;
    float* wA = icw + 2 * al4(rows); float* wU = wA + al4((size_t)rows * (N + 1));
    float* U = wU + al4((size_t)rows * (N + 1));
    float* W = U + al4(rows);                           // U and W are cleared together
    float* Cf = W + (size_t)rows * (grouped ? rpg4 : 0);
    float* G = Cf + (size_t)rows * 256;                  // [groups][rpg4][256]
    hipStream_t s = (hipStream_t)stream;
    if (grouped && hipMemsetAsync(U, 0, sizeof(float) * (al4(rows) + (size_t)rows * rpg4), s) != hipSuccess)
        return a8_set_error(A8_ERR_LAUNCH, "a8_contrastive_bwd: memset failed");
#define A8_CBWD(T)                                                                                                                     \
    {                                                                                                                                  \
        if (!inv_norms) hipLaunchKernelGGL(rownorm_kernel<T>, dim3(gridw(2 * rows)), dim3(256), 0, s, (const T*)ctx, (const T*)lat, icw, rows, Cn); \
        if (grouped) {                                                                                                                 \
            hipLaunchKernelGGL((contrastive_bwd_kernel<T, false>), dim3(gridw(rows)), dim3(256), 0, s, (const T*)ctx, (const T*)lat, neg_idx, ic, il, \
                               logits, gscale, dctx, dlat, wA, wU, rows, N);                                                           \
            const long long np = (long long)rows * (N + 1);                                                                            \
            hipLaunchKernelGGL(contrastive_w_kernel<T>, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, (const T*)ctx, (const T*)lat, neg_idx, \
                               il, wA, wU, W, U, dlat, rows, rpg, rpg4, N);                                                                  \
        } else {                                                                                                                       \
            hipLaunchKernelGGL((contrastive_bwd_kernel<T, true>), dim3(gridw(rows)), dim3(256), 0, s, (const T*)ctx, (const T*)lat, neg_idx, ic, il, \
                               logits, gscale, dctx, dlat, wA, wU, rows, N);                                                           \
        }                                                                                                                              \
    }
    if (dtype == A8_BF16) A8_CBWD(bf16) else A8_CBWD(float)
#undef A8_CBWD
    if (int e = a8_check_launch("a8_contrastive_bwd")) return e;
    if (!grouped) return A8_OK;
    // G_b [rpg, 256] = W_b^T [rpg x rpg] . C_b [rpg x 256] in exact f32, 8 utterances per grouped launch
    // G_b [rpg, 256] = W_b^T [rpg x rpg] . C_b [rpg x 256], 8 utterances per grouped launch: exact f32 in f32 mode; in
    // bf16 mode the coefficient matrix is rounded to bf16 (the result is rounded to bf16 by the caller anyway) and the
    // product runs on the bf16 MFMA straight from the bf16 context rows
    const void* amat = W;
    const void* bmat = ctx;
    if (dtype == A8_BF16) {
        if (int e = a8_cast(A8_F32, A8_BF16, W, Cf, (int64_t)rows * rpg4, stream)) return e;     // Cf: rows*256 floats >= rows*rpg4 bf16
        amat = Cf;
    }
    const int esz = dtype == A8_BF16 ? 2 : 4;
    const int ngrp = rows / rpg;
    for (int g0 = 0; g0 < ngrp; g0 += 8) {
        a8_gemm_desc d[8];
        const int cnt = ngrp - g0 < 8 ? ngrp - g0 : 8;
        for (int i = 0; i < cnt; ++i) {
            const size_t off = (size_t)(g0 + i) * rpg;
            d[i] = a8_gemm_desc{};
            d[i].size = (int)sizeof(a8_gemm_desc);
            d[i].dtype = dtype; d[i].layout = A8_GEMM_TN; d[i].M = rpg4; d[i].N = 256; d[i].K = rpg;
            d[i].A = (const char*)amat + off * rpg4 * esz; d[i].lda = rpg4;
            d[i].B = (const char*)bmat + off * 256 * esz; d[i].ldb = 256;
            d[i].C = G + (size_t)(g0 + i) * rpg4 * 256; d[i].ldc = 256;
            d[i].out_f32 = 1; d[i].splitk = 1; d[i].alpha = 1.0f;
        }
        if (int e = a8_gemm_grouped(d, cnt, stream)) return e;
    }
    const long long n = (long long)rows * 256;
    if (dtype == A8_BF16) hipLaunchKernelGGL(contrastive_combine_kernel<bf16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const bf16*)lat, il, G, U, dlat, n, rpg, rpg4);
    else hipLaunchKernelGGL(contrastive_combine_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)lat, il, G, U, dlat, n, rpg, rpg4);
    return a8_check_launch("a8_contrastive_bwd");
}

extern "C" int a8_loss_combine(const float* xe_sum, const float* ppl, int rows, float n_vars, float w_xe, float w_div, float* loss, void* stream) {
    A8_REQUIRE(rows > 0 && n_vars > 0.f, "a8_loss_combine: rows=%d and n_vars must be positive", rows);
    hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, xe_sum, ppl, (float)rows, n_vars, w_xe, w_div, loss);
    return a8_check_launch("a8_loss_combine");
}

extern "C" int a8_loss_combine_bwd(const float* dloss, int rows, float n_vars, float w_xe, float w_div, float* dxe_sum, float* dppl, void* stream) {
    A8_REQUIRE(rows > 0 && n_vars > 0.f, "a8_loss_combine_bwd: rows=%d and n_vars must be positive", rows);
    hipLaunchKernelGGL(loss_combine_bwd_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, dloss, (float)rows, n_vars, w_xe, w_div, dxe_sum, dppl);
    return a8_check_launch("a8_loss_combine_bwd");
}
