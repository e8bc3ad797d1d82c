// a8hip: conv layer 0 (Conv1d 1->512, k=10, s=5, no bias) + GroupNorm(512 groups = per-(b,c) instance norm
// over time) + exact GELU, forward and backward.   Reference: audio8/wav2vec2.py:417-423, 448-456.
//
// HBM-first design: the layer is 10 MACs per output, so instead of writing the pre-norm tensor
// (204.8 M elements at 8 x 250k samples) and re-reading it twice, every pass RECOMPUTES the conv from
// the waveform (1 MB per utterance, LDS-staged, coalesced along the sample axis):
//   forward  : stats pass (no output)  ->  finalize (mean, rstd per (b,c), f64)  ->  apply pass writes
//              gelu(gn(conv(x))) once, channels-last [B, L0, 512], 16 B per lane.
//   backward : ONE pass over dout accumulating per-(b,c) sums from which a tiny finalize kernel assembles
//              dw / dgamma / dbeta (the GroupNorm mean terms factor through channel-free waveform moments).
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

// ---- forward passes 1+2: GroupNorm statistics WITHOUT running the conv.  The conv is linear in 10 taps, so per (b,c)
//   sum_t v = sum_k w[c][k] S[k],   sum_t v^2 = sum_{j,k} w[c][j] w[c][k] R[j][k],
//   S[k] = sum_t x[5t+k],  R[j][k] = sum_t x[5t+j] x[5t+k]        (65 channel-free numbers per utterance)
// Pass 1 accumulates S and R per chunk of frames in f64 (products of f32 samples are exact in f64, so filters whose
// output is small against their taps' energy lose nothing to cancellation); pass 2 (one block per utterance) sums the
// chunks and evaluates the two forms per channel in f64.  (Was: a pass computing all 204.8 M conv outputs, 60 us.)
constexpr int CHF = 256;                         // frames per block of the apply pass
constexpr int CHM = 512;                         // frames per moments block
constexpr int NMOM = KW + KW * (KW + 1) / 2;     // S[10], then the upper triangle of R row by row
A8_DEV void mom_jk(int i, int& j, int& k) {      // moment index -> taps: i < 10: S[i] (k = -1); else R[j][k], j <= k, row by row
    if (i < KW) { j = i; k = -1; return; }
    int r = i - KW;
    j = 0;
    while (r >= KW - j) { r -= KW - j; ++j; }
    k = j + r;
}
// Lane l of every wave owns moment l; wave w walks frames w, w+4, ... over an f64 copy of the chunk (f64 converts and
// FMAs are quarter rate: convert once).  The 65th moment, R[9][9], is a thread-strided side loop.  No atomics.
__global__ __launch_bounds__(256) void conv0_moments_kernel(const float* __restrict__ x, double* __restrict__ part, int L, int L0, int nchunk) {
    __shared__ double xd[ST * CHM + 8];
    __shared__ double red[4][NMOM + 1];
    const int b = blockIdx.y, chunk = blockIdx.x, t0 = chunk * CHM;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    {
        const float* xb = x + (long long)b * L;
        const int s0 = ST * t0;
        for (int i = threadIdx.x; i < ST * CHM + (KW - ST); i += 256) xd[i] = (s0 + i < L) ? (double)xb[s0 + i] : 0.0;
    }
    __syncthreads();
    int j0, k0;
    mom_jk(lane, j0, k0);
    const int nfr = min(CHM, L0 - t0);
    double m0 = 0.0, m0b = 0.0, m1 = 0.0;
    // S moments multiply by a constant 1.0 parked behind the chunk (stride 0): one uniform loop for all lanes
    if (threadIdx.x == 0) xd[ST * CHM + 7] = 1.0;
    __syncthreads();
    const double* pj = xd + j0;
    const double* pk = k0 >= 0 ? xd + k0 : xd + ST * CHM + 7;
    const int sk = k0 >= 0 ? ST : 0;
    int tl = wv;
#pragma unroll 4
    for (; tl + 4 < nfr; tl += 8) {
        m0 = fma(pj[ST * tl], pk[sk * tl], m0);
        m0b = fma(pj[ST * (tl + 4)], pk[sk * (tl + 4)], m0b);
    }
    if (tl < nfr) m0 = fma(pj[ST * tl], pk[sk * tl], m0);
    m0 += m0b;
    for (int tl = threadIdx.x; tl < nfr; tl += 256) { const double e = xd[ST * tl + KW - 1]; m1 = fma(e, e, m1); }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m1 += __shfl_xor(m1, o, 64);
    red[wv][lane] = m0;
    if (lane == 0) red[wv][NMOM - 1] = m1;
    __syncthreads();
    if (threadIdx.x < NMOM)
        part[((long long)b * nchunk + chunk) * NMOM + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// one block of CC threads per utterance: (mean, rstd) per channel from the utterance's moments
__global__ __launch_bounds__(CC) void conv0_finalize_kernel(const double* __restrict__ part, const float* __restrict__ wg,
                                                            float* __restrict__ stats, int nchunk, int L0, float eps) {
    __shared__ double mom[NMOM];
    __shared__ double slab[7][NMOM];
    const int b = blockIdx.x, c = threadIdx.x;
    if (c < 7 * NMOM) {                          // 7 slices of the chunk list per moment
        const int i = c % NMOM, r = c / NMOM;
        double t = 0.0;
        for (int ch = r; ch < nchunk; ch += 7) t += part[((long long)b * nchunk + ch) * NMOM + i];
        slab[r][i] = t;
    }
    __syncthreads();
    if (c < NMOM) mom[c] = ((slab[0][c] + slab[1][c]) + (slab[2][c] + slab[3][c])) + ((slab[4][c] + slab[5][c]) + slab[6][c]);
    __syncthreads();
    double w[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) w[k] = (double)wg[c * KW + k];
    double s = 0.0, q = 0.0;
    int n = KW;
#pragma unroll
    for (int j = 0; j < KW; ++j) {
        s += w[j] * mom[j];
#pragma unroll
        for (int k = j; k < KW; ++k) { q += (j == k ? 1.0 : 2.0) * w[j] * w[k] * mom[n]; ++n; }
    }
    const double mu = s / L0;
    double var = q / L0 - mu * mu;
    if (var < 0.0) var = 0.0;
    stats[((long long)b * CC + c) * 2 + 0] = (float)mu;
    stats[((long long)b * CC + c) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
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
        for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], sc[j], sh[j]);
        gelu_n(v);
        st8(out + ((long long)b * L0 + t0 + tl) * CC + c0, v);
    }
}

// ---- forward pass 3 on the matrix cores (16-bit output).  The VALU form above spends 10 of its ~27 instructions per output on
// the taps and one on the GroupNorm affine; here both come out of ONE v_mfma_f32_16x16x32_bf16 per 16 frames x 16 channels:
//   u[c][t] = sum_k W'[c][k] X[k][t],   32 k-slots =  x_hi . w'_hi (10)  +  x_lo . w'_hi (10)  +  x_hi . w'_lo (10)  +  1 . sh_hi + 1 . sh_lo
// with w' = w * rstd * gamma and sh = beta - mean * rstd * gamma folded per (b, c), and every f32 operand split into two bf16
// (hi + lo, ~16 significand bits: the x_lo . w'_lo term that is dropped is 2^-18 of a product).  What is left for the VALU is
// the exact GELU and the pack.  Weights are the MFMA's row operand with the channels of one 64-channel super-block dealt
// c = 64 S + 16 (i >> 2) + 4 j + (i & 3) over the rows i of its four MFMAs j, so a lane ends up with 16 CONSECUTIVE channels of one
// frame (32 contiguous bytes of the channels-last row).
constexpr int MFR = 256;                          // frames per image build (one per thread)
// frames per block so that the grid is (just under) one round of `slots` co-resident blocks: a multiple of 16
static int conv0_frames_per_block(int B, int L0, int slots) {
    const int per_utt = slots / B > 0 ? slots / B : 1;
    const int fpb = (cdiv(L0, per_utt) + 15) / 16 * 16;
    return fpb < 16 ? 16 : fpb;
}
__global__ __launch_bounds__(256) void conv0_apply_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wg,
                                                               const float* __restrict__ gn_w, const float* __restrict__ gn_b,
                                                               const float* __restrict__ stats, bf16* __restrict__ out, int L, int L0, int fpb) {
    __shared__ __attribute__((aligned(16))) bf16 xim[MFR][32];     // per frame: the 32 k-slots of the x operand
    __shared__ __attribute__((aligned(16))) bf16 wim[CC][32];      // per channel: the 32 k-slots of the weight operand
    const int b = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, g = lane >> 4, li = lane & 15;
    auto split = [](float v, bf16& hi, bf16& lo) { hi = (bf16)v; lo = (bf16)(v - (float)hi); };
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {   // weight image of channels tid, tid + 256
        const int c = tid + 256 * cc;
        const float mu = stats[((long long)b * CC + c) * 2], rs = stats[((long long)b * CC + c) * 2 + 1];
        const float sc = rs * gn_w[c], sh = gn_b[c] - mu * sc;
        bf16 hi[KW], lo[KW], shh, shl;
#pragma unroll
        for (int k = 0; k < KW; ++k) split(wg[c * KW + k] * sc, hi[k], lo[k]);
        split(sh, shh, shl);
        bf16x8* row = reinterpret_cast<bf16x8*>(&wim[c][0]);
        row[0] = (bf16x8){hi[0], hi[1], hi[2], hi[3], hi[4], hi[5], hi[6], hi[7]};
        row[1] = (bf16x8){hi[8], hi[9], hi[0], hi[1], hi[2], hi[3], hi[4], hi[5]};
        row[2] = (bf16x8){hi[6], hi[7], hi[8], hi[9], lo[0], lo[1], lo[2], lo[3]};
        row[3] = (bf16x8){lo[4], lo[5], lo[6], lo[7], lo[8], lo[9], shh, shl};
    }
    const int tb = blockIdx.x * fpb, te = min(L0, tb + fpb);
    const float* xb = x + (long long)b * L;
    const int crow = 16 * (li >> 2) + (li & 3);                    // this lane's weight row within a super-block, + 4 j
    for (int t0 = tb; t0 < te; t0 += MFR) {
        __syncthreads();                                           // the previous x image is no longer read (and wim is complete)
        {   // x image of frame t0 + tid
            const int s0 = ST * (t0 + tid);
            bf16 hi[KW], lo[KW];
#pragma unroll
            for (int k = 0; k < KW; ++k) split(s0 + k < L ? xb[s0 + k] : 0.f, hi[k], lo[k]);
            const bf16 one = (bf16)1.0f;
            bf16x8* row = reinterpret_cast<bf16x8*>(&xim[tid][0]);
            row[0] = (bf16x8){hi[0], hi[1], hi[2], hi[3], hi[4], hi[5], hi[6], hi[7]};
            row[1] = (bf16x8){hi[8], hi[9], lo[0], lo[1], lo[2], lo[3], lo[4], lo[5]};
            row[2] = (bf16x8){lo[6], lo[7], lo[8], lo[9], hi[0], hi[1], hi[2], hi[3]};
            row[3] = (bf16x8){hi[4], hi[5], hi[6], hi[7], hi[8], hi[9], one, one};
        }
        __syncthreads();
        if (t0 + 64 * wv >= te) continue;                          // wave-uniform; barriers are outside this part
        // wave wv: frames t0 + 64 wv + 16 ft + li, ft = 0..3
        bf16x8 xf[4];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) xf[ft] = *reinterpret_cast<const bf16x8*>(&xim[64 * wv + 16 * ft + li][8 * g]);
        for (int S = 0; S < CC / 64; ++S) {
            bf16x8 wf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(&wim[64 * S + crow + 4 * j][8 * g]);
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) {
                f32x4 u[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) u[j] = Mma<bf16>::mma(wf[j], xf[ft], (f32x4){0.f, 0.f, 0.f, 0.f});
                const int t = t0 + 64 * wv + 16 * ft + li;
                bf16x8 o0, o1;                                     // channels 64 S + 16 g + 4 j + r
                float ua[8] = {u[0][0], u[0][1], u[0][2], u[0][3], u[1][0], u[1][1], u[1][2], u[1][3]};
                float ub[8] = {u[2][0], u[2][1], u[2][2], u[2][3], u[3][0], u[3][1], u[3][2], u[3][3]};
                gelu_n(ua);
                gelu_n(ub);
#pragma unroll
                for (int r = 0; r < 8; ++r) { o0[r] = (bf16)ua[r]; o1[r] = (bf16)ub[r]; }
                if (t < te) {
                    bf16* op = out + ((long long)b * L0 + t) * CC + 64 * S + 16 * g;
                    *reinterpret_cast<bf16x8*>(op) = o0;
                    *reinterpret_cast<bf16x8*>(op + 8) = o1;
                }
            }
        }
    }
}

// ---- backward: ONE pass over dout.
// With u = gamma*xh + beta, xh = (conv - mean)*rstd, dU = dout*gelu'(u), the GroupNorm backward is
//   dV = rstd*gamma*(dU - m1 - xh*m2),  m1 = mean_t dU, m2 = mean_t dU*xh,      dw[c][k] = sum_t dV[t] x[5t+k]
// and the sum splits into pieces that do not need m1, m2 in advance:
//   dw[c][k] = rstd*gamma*( A[c][k] - m1*S[k] - m2*C[c][k] ),
//   A[c][k] = sum_t dU x[5t+k]                      (accumulated here, per (b,c))
//   S[k]    = sum_t x[5t+k],  R[j][k] = sum_t x[5t+j] x[5t+k]     (waveform only, channel-free, f64)
//   C[c][k] = sum_t xh x[5t+k] = rstd*( sum_j w[c][j] R[j][k] - mean*S[k] )
// so dout (the 0.4 GB tensor) is read once; a finalize kernel over [B,512,10] assembles dw, dgamma, dbeta.
constexpr int CHB = 784;          // frames per block: 8 x 64 blocks x 4 waves = 2 waves per SIMD at 8 x 250k samples
constexpr int NACC = 2 + KW;      // per (b,c): s1, s2, A[0..9]
constexpr int NXR = KW + KW * (KW + 1) / 2;     // per b: S[10], then the upper triangle of R row by row (f64)
A8_DEV int tri(int j, int k) { return KW + j * KW - j * (j - 1) / 2 + (k - j); }     // j <= k

template <typename T>
__global__ __launch_bounds__(256) void conv0_bwd_main_kernel(const float* __restrict__ x, const float* __restrict__ wg,
                                                             const float* __restrict__ gn_w, const float* __restrict__ gn_b,
                                                             const float* __restrict__ stats, const T* __restrict__ dout,
                                                             float* __restrict__ acc, double* __restrict__ xr, int L, int L0) {
    __shared__ float xs[ST * CHB + 8];
    __shared__ float acc_s[CC * NACC];
    __shared__ float xr_s[NXR];
    const int b = blockIdx.y, t0 = blockIdx.x * CHB;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c0 = lane * 8;
    stage_x<CHB>(xs, x + (long long)b * L, t0, L);
    for (int i = threadIdx.x; i < CC * NACC; i += 256) acc_s[i] = 0.f;
    if (threadIdx.x < NXR) xr_s[threadIdx.x] = 0.f;
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
    const int nfr = min(CHB, L0 - t0);
    {   // waveform moments of this chunk: thread-strided frames, LDS atomics (65 values)
        float sx[KW], rx[KW * (KW + 1) / 2];
#pragma unroll
        for (int k = 0; k < KW; ++k) sx[k] = 0.f;
#pragma unroll
        for (int i = 0; i < KW * (KW + 1) / 2; ++i) rx[i] = 0.f;
        for (int tl = threadIdx.x; tl < nfr; tl += 256) {
            float xv[KW];
#pragma unroll
            for (int k = 0; k < KW; ++k) xv[k] = xs[ST * tl + k];
            int n = 0;
#pragma unroll
            for (int j = 0; j < KW; ++j) {
                sx[j] += xv[j];
#pragma unroll
                for (int k = j; k < KW; ++k) { rx[n] = fmaf(xv[j], xv[k], rx[n]); ++n; }
            }
        }
#pragma unroll
        for (int k = 0; k < KW; ++k) { const float t = wave_sum(sx[k]); if (lane == 0) atomicAdd(&xr_s[k], t); }
#pragma unroll
        for (int i = 0; i < KW * (KW + 1) / 2; ++i) { const float t = wave_sum(rx[i]); if (lane == 0) atomicAdd(&xr_s[KW + i], t); }
    }
    float s1[8], s2[8], a[8][KW];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        s1[j] = 0.f; s2[j] = 0.f;
#pragma unroll
        for (int k = 0; k < KW; ++k) a[j][k] = 0.f;
    }
    for (int tl = wv; tl < nfr; tl += 4) {
        float v[8], xv[KW], g[8];
        conv_frame(xs, tl, w, v, xv);
        ld8(dout + ((long long)b * L0 + t0 + tl) * CC + c0, g);
        float zz[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) zz[j] = fmaf((v[j] - mu[j]) * rs[j], gw[j], gb[j]);
        mul_gelu_grad_n(g, zz);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = (v[j] - mu[j]) * rs[j];
            const float du = g[j];
            s1[j] += du;
            s2[j] = fmaf(du, xh, s2[j]);
#pragma unroll
            for (int k = 0; k < KW; ++k) a[j][k] = fmaf(du, xv[k], a[j][k]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float* p = acc_s + (c0 + j) * NACC;
        atomicAdd(p, s1[j]);
        atomicAdd(p + 1, s2[j]);
#pragma unroll
        for (int k = 0; k < KW; ++k) atomicAdd(p + 2 + k, a[j][k]);
    }
    __syncthreads();
    float* ga = acc + (long long)b * CC * NACC;
    for (int i = threadIdx.x; i < CC * NACC; i += 256) atomicAdd(ga + i, acc_s[i]);
    if (threadIdx.x < NXR) atomicAdd(xr + (long long)b * NXR + threadIdx.x, (double)xr_s[threadIdx.x]);
}

// ---- the same backward pass on the matrix cores (16-bit dout).  Per 16 frames x 128 channels of a wave:
//   xh  = conv~ (taps * rstd, shift -mean * rstd folded: the forward's hi/lo-split product)       8 MFMA, frames on registers
//   dU  = dout * gelu'(gamma xh + beta);  s1 += dU;  s2 += dU xh                                    VALU (~29 instructions per element, was ~51)
//   A[c][k] += sum_frames dU[frame][c] x[5 frame + k]                                               8 MFMA: the dU accumulators ARE the operand
// Wave w owns channels 128 w .. 128 w + 127 (so per-(b,c) sums never cross waves); lane li of channel block j holds channel
// 128 w + 8 li + j, i.e. 8 consecutive channels across j: one 16-byte dout load per frame.
constexpr int BSUB = 256;                                        // frames per image rebuild
__global__ __launch_bounds__(256, 2) void conv0_bwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wg,
                                                             const float* __restrict__ gn_w, const float* __restrict__ gn_b,
                                                             const float* __restrict__ stats, const bf16* __restrict__ dout,
                                                             float* __restrict__ acc, double* __restrict__ xr, int L, int L0, int fpb) {
    __shared__ __attribute__((aligned(16))) float xs[ST * BSUB + 8];
    __shared__ __attribute__((aligned(16))) bf16 xim[BSUB][32];           // per frame: the 32 k-slots of the conv operand (as in the forward)
    __shared__ __attribute__((aligned(16))) bf16 x2im[BSUB / 16][64][8];  // per 16-frame step and lane: x[5 frame(slot) + tap], tap = lane & 15
    __shared__ __attribute__((aligned(16))) bf16 wim[CC][32];             // per channel: taps * rstd (hi, hi, lo), -mean * rstd (hi, lo)
    __shared__ float xr_s[NXR];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, g = lane >> 4, li = lane & 15;
    auto split = [](float v, bf16& hi, bf16& lo) { hi = (bf16)v; lo = (bf16)(v - (float)hi); };
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
        const int c = tid + 256 * cc;
        const float mu = stats[((long long)b * CC + c) * 2], rs = stats[((long long)b * CC + c) * 2 + 1];
        bf16 hi[KW], lo[KW], shh, shl;
#pragma unroll
        for (int k = 0; k < KW; ++k) split(wg[c * KW + k] * rs, hi[k], lo[k]);
        split(-mu * rs, shh, shl);
        bf16x8* row = reinterpret_cast<bf16x8*>(&wim[c][0]);
        row[0] = (bf16x8){hi[0], hi[1], hi[2], hi[3], hi[4], hi[5], hi[6], hi[7]};
        row[1] = (bf16x8){hi[8], hi[9], hi[0], hi[1], hi[2], hi[3], hi[4], hi[5]};
        row[2] = (bf16x8){hi[6], hi[7], hi[8], hi[9], lo[0], lo[1], lo[2], lo[3]};
        row[3] = (bf16x8){lo[4], lo[5], lo[6], lo[7], lo[8], lo[9], shh, shl};
    }
    if (tid < NXR) xr_s[tid] = 0.f;
    const int cbase = 128 * wv + 8 * li;                 // this lane's 8 channels (one per channel block j)
    float gw[8], gb[8], s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { gw[j] = gn_w[cbase + j]; gb[j] = gn_b[cbase + j]; s1[j] = 0.f; s2[j] = 0.f; }
    f32x4 accA[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) accA[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* xb = x + (long long)b * L;
    const int tb = blockIdx.x * fpb, te = min(L0, tb + fpb);
    for (int t0 = tb; t0 < te; t0 += BSUB) {             // block-uniform bounds
        __syncthreads();                                 // the previous images are no longer read
        {
            const int s0 = ST * t0;
            for (int i = tid; i < ST * BSUB + (KW - ST); i += 256) xs[i] = (s0 + i < L) ? xb[s0 + i] : 0.f;
        }
        __syncthreads();
        const int nfr = min(BSUB, te - t0);
        {   // images: conv operand of frame tid; tap operand of (step tid >> 6 and + 4, lane tid & 63)
            bf16 hi[KW], lo[KW];
#pragma unroll
            for (int k = 0; k < KW; ++k) split(xs[ST * tid + k], hi[k], lo[k]);
            const bf16 one = (bf16)1.0f;
            bf16x8* row = reinterpret_cast<bf16x8*>(&xim[tid][0]);
            row[0] = (bf16x8){hi[0], hi[1], hi[2], hi[3], hi[4], hi[5], hi[6], hi[7]};
            row[1] = (bf16x8){hi[8], hi[9], lo[0], lo[1], lo[2], lo[3], lo[4], lo[5]};
            row[2] = (bf16x8){lo[6], lo[7], lo[8], lo[9], hi[0], hi[1], hi[2], hi[3]};
            row[3] = (bf16x8){hi[4], hi[5], hi[6], hi[7], hi[8], hi[9], one, one};
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int st = (tid >> 6) + 4 * h, ln = tid & 63, gg = ln >> 4, tap = ln & 15;
                bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int e = 0; e < 4; ++e) {            // slot e < 4 of lane group gg <-> frame 4 gg + e of the step; slots 4 .. 7 stay zero
                    const int fr = 16 * st + 4 * gg + e;
                    v[e] = (tap < KW && fr < nfr) ? (bf16)xs[ST * fr + tap] : (bf16)0.f;
                }
                *reinterpret_cast<bf16x8*>(&x2im[st][ln][0]) = v;
            }
        }
        {   // waveform moments of these frames: thread-strided, LDS atomics (65 values)
            float sx[KW], rx[KW * (KW + 1) / 2];
#pragma unroll
            for (int k = 0; k < KW; ++k) sx[k] = 0.f;
#pragma unroll
            for (int i = 0; i < KW * (KW + 1) / 2; ++i) rx[i] = 0.f;
            if (tid < nfr) {
                float xv[KW];
#pragma unroll
                for (int k = 0; k < KW; ++k) xv[k] = xs[ST * tid + k];
                int n = 0;
#pragma unroll
                for (int j = 0; j < KW; ++j) {
                    sx[j] += xv[j];
#pragma unroll
                    for (int k = j; k < KW; ++k) { rx[n] = xv[j] * xv[k]; ++n; }
                }
            }
#pragma unroll
            for (int k = 0; k < KW; ++k) { const float t = wave_sum(sx[k]); if (lane == 0) atomicAdd(&xr_s[k], t); }
#pragma unroll
            for (int i = 0; i < KW * (KW + 1) / 2; ++i) { const float t = wave_sum(rx[i]); if (lane == 0) atomicAdd(&xr_s[KW + i], t); }
        }
        __syncthreads();
        bf16x8 wf[8];                                    // conv weights of this lane's 8 channels: the MFMA's column operand
#pragma unroll
        for (int j = 0; j < 8; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(&wim[cbase + j][8 * g]);
        const int nstep = (nfr + 15) / 16;
        // dout of frames 16 st + 4 g + r, channels cbase .. cbase + 7 (rows beyond the utterance read as zero); the next step's
        // rows are requested before this step's arithmetic starts
        auto load_dg = [&](int st, bf16x8 (&d)[4]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int fr = 16 * st + 4 * g + r;
                d[r] = fr < nfr ? *reinterpret_cast<const bf16x8*>(dout + ((long long)b * L0 + t0 + fr) * CC + cbase) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
            }
        };
        bf16x8 dg[4], dn[4];
        load_dg(0, dg);
        for (int st = 0; st < nstep; ++st) {
            if (st + 1 < nstep) load_dg(st + 1, dn);
            const bf16x8 xf0 = *reinterpret_cast<const bf16x8*>(&xim[16 * st + li][8 * g]);
            const bf16x8 x2 = *reinterpret_cast<const bf16x8*>(&x2im[st][lane][0]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // xh[frame 4 g + r][channel cbase + j]: frames are the MFMA's rows, channels its columns
                const f32x4 h0 = Mma<bf16>::mma(xf0, wf[j], (f32x4){0.f, 0.f, 0.f, 0.f});
                bf16x8 du = {0, 0, 0, 0, 0, 0, 0, 0};
                float dd[4], zz[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { dd[r] = (float)dg[r][j]; zz[r] = fmaf(h0[r], gw[j], gb[j]); }
                mul_gelu_grad_n(dd, zz);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s1[j] += dd[r];
                    s2[j] = fmaf(dd[r], h0[r], s2[j]);
                    du[r] = (bf16)dd[r];
                }
                // A[channel 8 i + j][tap] += sum over the 16 frames (k-slots 4 .. 7 of every lane group are zero on both sides)
                accA[j] = Mma<bf16>::mma(du, x2, accA[j]);
                if (j & 1) __builtin_amdgcn_sched_barrier(0);      // two channel blocks in flight at a time: bounds the live registers
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) dg[r] = dn[r];
        }
    }
    // ---- per-(b,c) sums out: s1, s2 over the 4 lane groups (same li = same channels), A from the accumulators
    float* ga = acc + (long long)b * CC * NACC;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float a = s1[j], q = s2[j];
        a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
        q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
        if (g == 0) { atomicAdd(ga + (cbase + j) * NACC, a); atomicAdd(ga + (cbase + j) * NACC + 1, q); }
        if (li < KW) {                                   // accA[j][r]: channel 128 wv + 8 (4 g + r) + j, tap li
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(ga + (128 * wv + 8 * (4 * g + r) + j) * NACC + 2 + li, accA[j][r]);
        }
    }
    __syncthreads();
    if (tid < NXR) atomicAdd(xr + (long long)b * NXR + tid, (double)xr_s[tid]);
}

// ---- finalize: dw[c][k] += sum_b rstd*gamma*(A - m1*S - m2*C); dgn_w[c] += sum_b s2; dgn_b[c] += sum_b s1
__global__ void conv0_bwd_finalize_kernel(const float* __restrict__ wg, const float* __restrict__ gn_w, const float* __restrict__ stats,
                                          const float* __restrict__ acc, const double* __restrict__ xr,
                                          float* __restrict__ dw, float* __restrict__ dgn_w, float* __restrict__ dgn_b, int B, int L0) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= CC) return;
    float wc[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) wc[k] = wg[c * KW + k];
    const double invL = 1.0 / (double)L0;
    double dwk[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) dwk[k] = 0.0;
    double dgw = 0.0, dgb = 0.0;
    for (int b = 0; b < B; ++b) {
        const float* ab = acc + ((long long)b * CC + c) * NACC;
        const double* xb = xr + (long long)b * NXR;
        const double mu = stats[((long long)b * CC + c) * 2], rs = stats[((long long)b * CC + c) * 2 + 1];
        const double s1 = ab[0], s2 = ab[1], m1 = s1 * invL, m2 = s2 * invL;
        dgb += s1;
        dgw += s2;
        const double f = rs * (double)gn_w[c];
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            double wr = 0.0;
#pragma unroll
            for (int j = 0; j < KW; ++j) wr += (double)wc[j] * xb[j <= k ? tri(j, k) : tri(k, j)];
            const double ck = rs * (wr - mu * xb[k]);
            dwk[k] += f * ((double)ab[2 + k] - m1 * xb[k] - m2 * ck);
        }
    }
#pragma unroll
    for (int k = 0; k < KW; ++k) dw[c * KW + k] += (float)dwk[k];
    dgn_w[c] += (float)dgw;
    dgn_b[c] += (float)dgb;
}

}  // namespace

static int conv0_out_len(int L) { return (L - KW) / ST + 1; }

extern "C" int a8_conv0_gn_gelu_fwd(int dtype, const float* x, const float* w, const float* gn_w, const float* gn_b,
                                    void* out, float* stats, float* part, int B, int L, int C_, float eps, void* stream) {
    A8_REQUIRE(C_ == CC, "a8_conv0_gn_gelu_fwd: C must be %d (got %d)", CC, C_);
    A8_REQUIRE(L >= KW && B >= 1, "a8_conv0_gn_gelu_fwd: bad B=%d L=%d", B, L);
    A8_REQUIRE(dtype == A8_F32 || dtype == A8_BF16, "a8_conv0_gn_gelu_fwd: bad dtype");
    A8_REQUIRE(((uintptr_t)part % 8) == 0, "a8_conv0_gn_gelu_fwd: scratch must be 8-byte aligned");
    const int L0 = conv0_out_len(L), nchunk = cdiv(L0, CHF), nmom = cdiv(L0, CHM);
    hipStream_t s = (hipStream_t)stream;
    double* mpart = reinterpret_cast<double*>(part);          // [B][nmom][65] doubles
    hipLaunchKernelGGL(conv0_moments_kernel, dim3(nmom, B), dim3(256), 0, s, x, mpart, L, L0, nmom);
    hipLaunchKernelGGL(conv0_finalize_kernel, dim3(B), dim3(CC), 0, s, (const double*)mpart, w, stats, nmom, L0, eps);
    if (dtype == A8_BF16 && g_a8_tuning[9] == 0) {     // (a8_set_tuning key 9 = 1: the VALU form, for A/B runs)
        const int fpb = g_a8_tuning[10] > 0 ? g_a8_tuning[10] : MFR;   // one image build per block measured best (1568 small blocks at cfg2:
                                                                       // 133-138 us; 760 blocks sized to one resident round: 151 us)
        hipLaunchKernelGGL(conv0_apply_mfma_kernel, dim3(cdiv(L0, fpb), B), dim3(256), 0, s, x, w, gn_w, gn_b, stats, (bf16*)out, L, L0, fpb);
    }
    else if (dtype == A8_BF16)
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
    A8_REQUIRE(((uintptr_t)red % 8) == 0, "a8_conv0_gn_gelu_bwd: scratch must be 8-byte aligned");
    const int L0 = conv0_out_len(L), nchunk = cdiv(L0, CHB);
    hipStream_t s = (hipStream_t)stream;
    // scratch layout: f64 xr[B][NXR] | f32 acc[B][C][NACC]   (A8_CONV0_BWD_SCRATCH_FLOATS(B) floats in all)
    double* xr = reinterpret_cast<double*>(red);
    float* acc = red + 2 * (size_t)B * NXR;
    static_assert(NXR == 65 && NACC == 12, "keep A8_CONV0_BWD_SCRATCH_FLOATS in a8hip.h in sync");
    if (hipMemsetAsync(red, 0, sizeof(float) * A8_CONV0_BWD_SCRATCH_FLOATS(B), s) != hipSuccess)
        return a8_set_error(A8_ERR_LAUNCH, "a8_conv0_gn_gelu_bwd: memset failed");
    if (dtype == A8_BF16 && g_a8_tuning[9] == 0) {
        const int fpb = conv0_frames_per_block(B, L0, 512);        // 256 registers: two blocks per CU
        hipLaunchKernelGGL(conv0_bwd_mfma_kernel, dim3(cdiv(L0, fpb), B), dim3(256), 0, s, x, w, gn_w, gn_b, stats, (const bf16*)dout, acc, xr, L, L0, fpb);
    }
    else if (dtype == A8_BF16)
        hipLaunchKernelGGL(conv0_bwd_main_kernel<bf16>, dim3(nchunk, B), dim3(256), 0, s, x, w, gn_w, gn_b, stats, (const bf16*)dout, acc, xr, L, L0);
    else
        hipLaunchKernelGGL(conv0_bwd_main_kernel<float>, dim3(nchunk, B), dim3(256), 0, s, x, w, gn_w, gn_b, stats, (const float*)dout, acc, xr, L, L0);
    hipLaunchKernelGGL(conv0_bwd_finalize_kernel, dim3(cdiv(CC, 64)), dim3(64), 0, s, w, gn_w, stats, acc, xr, dw, dgn_w, dgn_b, B, L0);
    return a8_check_launch("a8_conv0_gn_gelu_bwd");
}
