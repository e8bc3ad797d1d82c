// a8hip: convolutional positional embedding (grouped Conv1d(d, d, k=128, groups=16) + bias + GELU, "same"
// padding) of AudioTransformerEncoder -- audio8/wav2vec2.py:599-609 (ctor), 634-638 (use).
//
// Implicit GEMM per group on MFMA: out[t][co] = sum_{kk,ci} xpad[t+kk][ci] * W[co][kk][ci].
// A block owns 256 frames of one (utterance, group): the 256+127-row sliding window of the group's dg
// channels is staged ONCE in LDS and every tap kk reads it at a row offset (no im2col anywhere); the
// group's weights stream through LDS in 512-byte-per-row chunks (register-prefetched), shared by 8 waves.
// The data gradient is the same kernel run on dz with flipped/transposed weights and the pads swapped.
// The weight gradient contracts over time with both operands read transposed from LDS.
#include "a8_common.h"
#include "a8hip_internal.h"

namespace {

constexpr int KW = 128;
constexpr int BM = 256;

struct PosP {
    const char *x, *w, *res;
    const float* bias;
    char *y, *z;
    int B, T, d, pad_left, add_residual, apply_gelu;
};

template <typename T, int DG> struct PCfg {
    static constexpr int EPC = 16 / (int)sizeof(T);
    static constexpr int CPX = DG / EPC;                        // 16-byte chunks per window row
    static constexpr int XROW = DG * (int)sizeof(T) + 16;       // padded window row (bytes)
    static constexpr int KCH = 512 / (int)sizeof(T);            // reduction elements per weight chunk
    static constexpr int NB = DG / 16;
    static constexpr int WIN = BM + KW - 1;
    static constexpr int NCHUNK = KW * DG / KCH;
    static constexpr int WPT = DG * 32 / 512;                   // weight 16-byte chunks per thread per chunk
};

template <typename T, int DG>
__global__ __launch_bounds__(512) void posconv_fwd_kernel(const PosP P) {
    typedef PCfg<T, DG> Cf;
    typedef Mma<T> MM;
    __shared__ __attribute__((aligned(16))) char xwin[Cf::WIN * Cf::XROW];
    __shared__ __attribute__((aligned(16))) char wt[DG * 512];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g4 = lane >> 4, li = lane & 15;
    const int grp = blockIdx.y, b = blockIdx.z, t0 = blockIdx.x * BM;
    const T* xb = reinterpret_cast<const T*>(P.x) + (long long)b * P.T * P.d + grp * DG;
    for (int c = tid; c < Cf::WIN * Cf::CPX; c += 512) {
        const int r = c / Cf::CPX, cc = c % Cf::CPX;
        const int ts = t0 + r - P.pad_left;
        const u32x4 v = (ts >= 0 && ts < P.T) ? ld16(xb + (long long)ts * P.d + cc * Cf::EPC) : (u32x4){0u, 0u, 0u, 0u};
        st16(xwin + r * Cf::XROW + cc * 16, v);
    }
    const T* wg = reinterpret_cast<const T*>(P.w) + (long long)grp * DG * (KW * DG);
    u32x4 wr[Cf::WPT];
    auto wload = [&](int ch) {
#pragma unroll
        for (int j = 0; j < Cf::WPT; ++j) {
            const int idx = tid + 512 * j, co = idx >> 5, c16 = idx & 31;
            wr[j] = ld16(wg + (long long)co * (KW * DG) + ch * Cf::KCH + c16 * Cf::EPC);
        }
    };
    auto wstore = [&]() {
#pragma unroll
        for (int j = 0; j < Cf::WPT; ++j) {
            const int idx = tid + 512 * j, co = idx >> 5, c16 = idx & 31;
            st16(wt + co * 512 + ((c16 ^ (co & 15)) << 4), wr[j]);
        }
    };
    f32x4 acc[2][Cf::NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < Cf::NB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    wload(0);
    wstore();
    __syncthreads();
    for (int ch = 0; ch < Cf::NCHUNK; ++ch) {
        if (ch + 1 < Cf::NCHUNK) wload(ch + 1);
#pragma unroll 4
        for (int ks = 0; ks < Cf::KCH / MM::KI; ++ks) {
            typename MM::Frag xa[2], wb[Cf::NB];
            if constexpr (sizeof(T) == 2) {
                const int chi = (ch * Cf::KCH + ks * 32) / 8 + g4;          // 16-byte chunk index along kappa
                const int kk = chi / Cf::CPX, cic = chi % Cf::CPX;
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    xa[mi] = *reinterpret_cast<const bf16x8*>(xwin + (wv * 32 + 16 * mi + li + kk) * Cf::XROW + cic * 16);
#pragma unroll
                for (int nb = 0; nb < Cf::NB; ++nb) {
                    const int co = 16 * nb + li;
                    wb[nb] = *reinterpret_cast<const bf16x8*>(wt + co * 512 + (((ks * 4 + g4) ^ (co & 15)) << 4));
                }
            } else {
                const int kap = ch * Cf::KCH + ks * 4 + g4;
                const int kk = kap / DG, ci = kap % DG;
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    xa[mi] = *reinterpret_cast<const float*>(xwin + (wv * 32 + 16 * mi + li + kk) * Cf::XROW + ci * 4);
#pragma unroll
                for (int nb = 0; nb < Cf::NB; ++nb) {
                    const int co = 16 * nb + li;
                    wb[nb] = *reinterpret_cast<const float*>(wt + co * 512 + ((ks ^ (co & 15)) << 4) + g4 * 4);
                }
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int nb = 0; nb < Cf::NB; ++nb) acc[mi][nb] = MM::mma(wb[nb], xa[mi], acc[mi][nb]);
        }
        __syncthreads();
        if (ch + 1 < Cf::NCHUNK) wstore();
        __syncthreads();
    }
    // epilogue: lane holds 4 consecutive output channels of frame t
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int t = t0 + wv * 32 + 16 * mi + li;
        if (t >= P.T) continue;
#pragma unroll
        for (int nb = 0; nb < Cf::NB; ++nb) {
            const int c = grp * DG + 16 * nb + 4 * g4;
            const long long off = ((long long)b * P.T + t) * P.d + c;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[mi][nb][r] + (P.bias ? P.bias[c + r] : 0.f);
            if (P.z) {
                T* zp = reinterpret_cast<T*>(P.z) + off;
#pragma unroll
                for (int r = 0; r < 4; ++r) zp[r] = from_f<T>(v[r]);
            }
            if (P.apply_gelu) gelu_n(v);
            if (P.add_residual) {
                const T* rp = reinterpret_cast<const T*>(P.res) + off;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += to_f(rp[r]);
            }
            T* yp = reinterpret_cast<T*>(P.y) + off;
#pragma unroll
            for (int r = 0; r < 4; ++r) yp[r] = from_f<T>(v[r]);
        }
    }
}

// ------------------------------------------------------------------ weight gradient
// dw[grp][co][kk][ci] += sum_{b,t} dz[b,t,grp*DG+co] * xpad[b,t+kk,grp*DG+ci]
// block = (group, 8 taps); wave = 2 taps; loops over (b, 64-frame chunks); both operands via transposed LDS reads.
constexpr int WG_TAPS = 8, WG_BT = 64;
template <typename T, int DG>
__global__ __launch_bounds__(256) void posconv_wgrad_kernel(const char* __restrict__ xp, const char* __restrict__ dzp, float* __restrict__ dw,
                                                            int B, int Tn, int d, int pad_left) {
    constexpr int EPC = 16 / (int)sizeof(T), CPX = DG / EPC, NB = DG / 16;
    constexpr int XR = WG_BT + WG_TAPS - 1;
    // LDS image of a window row.  16-bit data is read back transposed (ds_read_b64_tr_b16: the 32 lanes of a group take 32 contiguous bytes of
    // rows R .. R+3 and R+8 .. R+11, banks (address / 4) mod 64): with rows 96 B (DG = 48) or 160 B (DG = 64) apart and every block of 8 rows
    // shifted by another 128 B those eight pieces cover the 64 banks exactly once for every R and column block.  (Round 6: the padded-row image
    // [row][DG * 2 + 16] was 2-way conflicted on every read, 10.4 M conflict cycles per launch in profiles/r05_kernel_sq_counters.txt.)
    constexpr bool TR = sizeof(T) == 2;
    constexpr int XROW = TR ? (DG == 48 ? 96 : DG * 2 + 32) : DG * (int)sizeof(T) + 16;
    auto roff = [](int r) { return TR ? r * XROW + 128 * (r >> 3) : r * XROW; };
    constexpr int XS_BYTES = XR * XROW + (TR ? 128 * ((XR + 7) / 8) : 0), ZS_BYTES = WG_BT * XROW + (TR ? 128 * (WG_BT / 8) : 0);
    __shared__ __attribute__((aligned(16))) char xs[XS_BYTES];
    __shared__ __attribute__((aligned(16))) char zs[ZS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, g4 = lane >> 4, li = lane & 15;
    const int grp = blockIdx.y, kk0 = blockIdx.x * WG_TAPS;
    f32x4 acc[2][NB][NB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[a][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nt = (Tn + WG_BT - 1) / WG_BT;
    // blockIdx.z splits the (utterance, 64-frame chunk) loop: 256 (group, tap-block) pairs alone leave one 4-wave block per CU
    const int per = (B * nt + (int)gridDim.z - 1) / (int)gridDim.z;
    const int it_end = min(B * nt, ((int)blockIdx.z + 1) * per);
    // The operands of iteration it + 1 are requested into registers before iteration it's MFMAs (round 6: the loop was load -> store -> barrier ->
    // compute, one exposed global-memory round trip per 64 frames: ~35 of them per block were the kernel's 115 us).
    constexpr int XL = (XR * CPX + 255) / 256, ZL = (WG_BT * CPX + 255) / 256;
    u32x4 xr[XL], zr[ZL];
    auto request = [&](int it) {
        const int b = it / nt, t0 = (it % nt) * WG_BT;
        const T* xb = reinterpret_cast<const T*>(xp) + (long long)b * Tn * d + grp * DG;
        const T* zb = reinterpret_cast<const T*>(dzp) + (long long)b * Tn * d + grp * DG;
#pragma unroll
        for (int j = 0; j < XL; ++j) {
            const int c = tid + 256 * j, r = c / CPX, cc = c % CPX, ts = t0 + r + kk0 - pad_left;
            xr[j] = (c < XR * CPX && ts >= 0 && ts < Tn) ? ld16(xb + (long long)ts * d + cc * EPC) : (u32x4){0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int j = 0; j < ZL; ++j) {
            const int c = tid + 256 * j, r = c / CPX, cc = c % CPX, ts = t0 + r;
            zr[j] = (c < WG_BT * CPX && ts < Tn) ? ld16(zb + (long long)ts * d + cc * EPC) : (u32x4){0u, 0u, 0u, 0u};
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < XL; ++j) {
            const int c = tid + 256 * j, r = c / CPX, cc = c % CPX;
            if (c < XR * CPX) st16(xs + roff(r) + cc * 16, xr[j]);
        }
#pragma unroll
        for (int j = 0; j < ZL; ++j) {
            const int c = tid + 256 * j, r = c / CPX, cc = c % CPX;
            if (c < WG_BT * CPX) st16(zs + roff(r) + cc * 16, zr[j]);
        }
    };
    const int it_beg = blockIdx.z * per;
    if (it_beg < it_end) request(it_beg);
    for (int it = it_beg; it < it_end; ++it) {
        __syncthreads();                              // every wave is done reading the previous iteration's image
        commit();
        __syncthreads();
        if (it + 1 < it_end) request(it + 1);
        if constexpr (sizeof(T) == 2) {
            const int q = li >> 2, p = li & 3;
#pragma unroll
            for (int ks = 0; ks < WG_BT / 32; ++ks) {
                bf16x8 za[NB];
                const int r0 = ks * 32 + 8 * g4 + q;
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) {
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, zs + roff(r0) + (16 * cb + 4 * p) * 2));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, zs + roff(r0 + 4) + (16 * cb + 4 * p) * 2));
                    typedef short s16x8 __attribute__((ext_vector_type(8)));
                    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    za[cb] = __builtin_bit_cast(bf16x8, v);
                }
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int kk = wv * 2 + a;
#pragma unroll
                    for (int ib = 0; ib < NB; ++ib) {
                        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, xs + roff(r0 + kk) + (16 * ib + 4 * p) * 2));
                        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, xs + roff(r0 + kk + 4) + (16 * ib + 4 * p) * 2));
                        typedef short s16x8 __attribute__((ext_vector_type(8)));
                        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        const bf16x8 xbv = __builtin_bit_cast(bf16x8, v);
#pragma unroll
                        for (int cb = 0; cb < NB; ++cb) acc[a][cb][ib] = Mma<bf16>::mma(za[cb], xbv, acc[a][cb][ib]);
                    }
                }
            }
        } else {
#pragma unroll 4
            for (int ks = 0; ks < WG_BT / 4; ++ks) {
                float za[NB];
                const int r0 = ks * 4 + g4;
#pragma unroll
                for (int cb = 0; cb < NB; ++cb) za[cb] = *reinterpret_cast<const float*>(zs + roff(r0) + (16 * cb + li) * 4);
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int kk = wv * 2 + a;
#pragma unroll
                    for (int ib = 0; ib < NB; ++ib) {
                        const float xbv = *reinterpret_cast<const float*>(xs + roff(r0 + kk) + (16 * ib + li) * 4);
#pragma unroll
                        for (int cb = 0; cb < NB; ++cb) acc[a][cb][ib] = Mma<float>::mma(za[cb], xbv, acc[a][cb][ib]);
                    }
                }
            }
        }
    }
    // acc[a][cb][ib][r] = dW[co = 16cb + 4g4 + r][kk][ci = 16ib + li]
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int kk = kk0 + wv * 2 + a;
#pragma unroll
        for (int cb = 0; cb < NB; ++cb)
#pragma unroll
            for (int ib = 0; ib < NB; ++ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = 16 * cb + 4 * g4 + r, ci = 16 * ib + li;
                    atomicAdd(dw + (((long long)grp * DG + co) * KW + kk) * DG + ci, acc[a][cb][ib][r]);
                }
    }
}

// ------------------------------------------------------------------ weight norm (nn.utils.weight_norm(dim=2))
// v f32 [d][dg][k]; w[o][ci][kk] = g[kk] * v[o][ci][kk] / ||v[:, :, kk]||  (audio8/wav2vec2.py:609).
// Three small launches each way instead of the tape's dozen element-wise passes over the 4.7 M-element kernel:
// per-output-channel partial sums -> a few reducing blocks -> a transposing pass that writes the operand images of
// posconv_fwd_kernel (forward) or accumulates dv (backward).  Slabs go through a padded LDS tile so that both the
// global read and the global write are contiguous.  DG / KK = 0: run-time sizes (any k that is a power of two in 16..128).
constexpr int WN_MAXK = 128, WN_MAXDG = 64;
template <int DG, int KK> struct WnDims {
    int dg_, k_;
    A8_DEV WnDims(int dg, int k) : dg_(dg), k_(k) {}
    A8_DEV int dg() const { return DG ? DG : dg_; }
    A8_DEV int k() const { return KK ? KK : k_; }
};

// part[o][kk] = sum_ci v[o][ci][kk]^2                              (dwg == nullptr)
//             = sum_ci dwg[o][kk][ci] * v[o][ci][kk]               (backward: dwg is [d][k][dg])
template <int DG, int KK>
__global__ __launch_bounds__(256) void wn_partial_kernel(const float* __restrict__ v, const float* __restrict__ dwg,
                                                         float* __restrict__ part, int dg_rt, int k_rt) {
    __shared__ float tile[WN_MAXK * (WN_MAXDG + 1)];
    __shared__ float red[256];
    const WnDims<DG, KK> D(dg_rt, k_rt);
    const int dg = D.dg(), k = D.k();
    const int o = blockIdx.x, n = dg * k;
    const float* vo = v + (size_t)o * n;
    if (dwg) {
        const float* dwo = dwg + (size_t)o * n;
#pragma unroll 4
        for (int e = threadIdx.x; e < n; e += 256) tile[(e / dg) * (dg + 1) + e % dg] = dwo[e];
        __syncthreads();
    }
    const int rows = 256 / k, r = threadIdx.x / k, kk = threadIdx.x % k;         // k divides 256
    float acc = 0.f;
#pragma unroll 8
    for (int ci = r; ci < dg; ci += rows) {
        const float x = vo[ci * k + kk];
        acc += dwg ? tile[kk * (dg + 1) + ci] * x : x * x;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (r == 0) {
        for (int j = 1; j < rows; ++j) acc += red[j * k + kk];
        part[(size_t)o * k + kk] = acc;
    }
}

// Block b owns columns [16 b, 16 b + 16): 64 row groups x 16 columns.
// fwd (dg_out == nullptr): stat[kk] = 1 / sqrt(sum_o part[o][kk])            (inverse norm)
// bwd: stat[kk] = dot[kk] = sum_o part[o][kk];  dg_out[kk] += dot[kk] * inv_norm[kk]
__global__ __launch_bounds__(1024) void wn_reduce_kernel(const float* __restrict__ part, const float* __restrict__ inv_norm,
                                                         float* __restrict__ stat, float* __restrict__ dg_out, int d, int k) {
    __shared__ float red[1024];
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15, kk = blockIdx.x * 16 + c;
    float acc = 0.f;
#pragma unroll 4
    for (int o = r; o < d; o += 64) acc += part[(size_t)o * k + kk];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int h = 32; h >= 1; h >>= 1) {
        if (r < h) red[threadIdx.x] += red[threadIdx.x + 16 * h];
        __syncthreads();
    }
    if (r == 0) {
        acc = red[c];
        if (dg_out) {
            stat[kk] = acc;
            dg_out[kk] += acc * inv_norm[kk];
        } else {
            stat[kk] = 1.0f / sqrtf(acc);
        }
    }
}

// blockIdx.y = 0: wf[o][kk][ci] = s[kk] v[o][ci][kk]                 (o = g*dg + co: [G][co][kk][ci])
// blockIdx.y = 1: block b = (g, ci): wt[g][ci][kk'][co] = s[k-1-kk'] v[g*dg+co][ci][k-1-kk'],  s = g * inv_norm
template <typename T, int DG, int KK>
__global__ __launch_bounds__(256) void wn_pack_kernel(const float* __restrict__ v, const float* __restrict__ gain,
                                                      const float* __restrict__ inv_norm, T* __restrict__ wf, T* __restrict__ wt,
                                                      int dg_rt, int k_rt) {
    __shared__ float tile[WN_MAXDG * (WN_MAXK + 1)];
    __shared__ float sc[WN_MAXK];
    const WnDims<DG, KK> D(dg_rt, k_rt);
    const int dg = D.dg(), k = D.k();
    const int b = blockIdx.x, n = dg * k;
    for (int kk = threadIdx.x; kk < k; kk += 256) sc[kk] = gain[kk] * inv_norm[kk];
    if (blockIdx.y == 0) {
        const float* vo = v + (size_t)b * n;
#pragma unroll 4
        for (int e = threadIdx.x; e < n; e += 256) tile[(e / k) * (k + 1) + e % k] = vo[e];
        __syncthreads();
        T* dst = wf + (size_t)b * n;
#pragma unroll 4
        for (int e = threadIdx.x; e < n; e += 256) {
            const int kk = e / dg, ci = e % dg;
            dst[e] = from_f<T>(tile[ci * (k + 1) + kk] * sc[kk]);
        }
    } else {
        const int g = b / dg, ci = b % dg;
#pragma unroll 4
        for (int e = threadIdx.x; e < n; e += 256) {
            const int co = e / k, kk = e % k;
            tile[co * (k + 1) + kk] = v[((size_t)(g * dg + co) * dg + ci) * k + kk];
        }
        __syncthreads();
        T* dst = wt + (size_t)b * n;
#pragma unroll 4
        for (int e = threadIdx.x; e < n; e += 256) {
            const int kf = k - 1 - e / dg, co = e % dg;
            dst[e] = from_f<T>(tile[co * (k + 1) + kf] * sc[kf]);
        }
    }
}

// dv[o][ci][kk] += s[kk] * (dwg[o][kk][ci] - v[o][ci][kk] * dot[kk] * inv_norm[kk]^2)
template <int DG, int KK>
__global__ __launch_bounds__(256) void wn_bwd_kernel(const float* __restrict__ dwg, const float* __restrict__ v,
                                                     const float* __restrict__ gain, const float* __restrict__ inv_norm,
                                                     const float* __restrict__ dot, float* __restrict__ dv, int dg_rt, int k_rt) {
    __shared__ float tile[WN_MAXK * (WN_MAXDG + 1)];
    __shared__ float sc[WN_MAXK], pj[WN_MAXK];
    const WnDims<DG, KK> D(dg_rt, k_rt);
    const int dg = D.dg(), k = D.k();
    const int o = blockIdx.x, n = dg * k;
    for (int kk = threadIdx.x; kk < k; kk += 256) {
        const float in = inv_norm[kk];
        sc[kk] = gain[kk] * in;
        pj[kk] = dot[kk] * in * in;
    }
    const float* dwo = dwg + (size_t)o * n;
#pragma unroll 4
    for (int e = threadIdx.x; e < n; e += 256) tile[(e / dg) * (dg + 1) + e % dg] = dwo[e];
    __syncthreads();
    const float* vo = v + (size_t)o * n;
    float* dvo = dv + (size_t)o * n;
#pragma unroll 4
    for (int e = threadIdx.x; e < n; e += 256) {
        const int ci = e / k, kk = e % k;
        dvo[e] += sc[kk] * (tile[kk * (dg + 1) + ci] - vo[e] * pj[kk]);
    }
}

// compile-time sizes for the two shapes audio8 builds (d/groups = 48 or 64, k = 128); anything else takes run-time sizes
#define WN_DISPATCH(dg, k, CALL) do { if ((k) == 128 && (dg) == 48) { CALL(48, 128); } else if ((k) == 128 && (dg) == 64) { CALL(64, 128); } else { CALL(0, 0); } } while (0)

}  // namespace

extern "C" int a8_posconv_fwd(int dtype, const void* x, const void* w, const float* bias, void* y, void* z,
                              int B, int T_, int d, int groups, int k, int pad_left, int add_residual, void* stream) {
    A8_REQUIRE(k == KW, "a8_posconv_fwd: kernel width must be %d (got %d)", KW, k);
    A8_REQUIRE(groups >= 1 && d % groups == 0, "a8_posconv_fwd: d=%d not divisible by groups=%d", d, groups);
    const int dg = d / groups;
    A8_REQUIRE(dg == 48 || dg == 64, "a8_posconv_fwd: channels per group must be 48 or 64 (got %d)", dg);
    A8_REQUIRE(pad_left >= 0 && pad_left < KW, "a8_posconv_fwd: bad pad_left");
    A8_REQUIRE(dtype == A8_F32 || dtype == A8_BF16, "a8_posconv_fwd: bad dtype");
    PosP P = {};
    P.x = (const char*)x; P.w = (const char*)w; P.res = (const char*)x; P.bias = bias; P.y = (char*)y; P.z = (char*)z;
    P.B = B; P.T = T_; P.d = d; P.pad_left = pad_left; P.add_residual = add_residual; P.apply_gelu = add_residual >= 0 ? 1 : 0;
    if (add_residual < 0) { P.add_residual = 0; }      // add_residual = -1: plain conv (data-gradient mode)
    const dim3 grid(cdiv(T_, BM), groups, B);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == A8_BF16) {
        if (dg == 48) hipLaunchKernelGGL((posconv_fwd_kernel<bf16, 48>), grid, dim3(512), 0, s, P);
        else hipLaunchKernelGGL((posconv_fwd_kernel<bf16, 64>), grid, dim3(512), 0, s, P);
    } else {
        if (dg == 48) hipLaunchKernelGGL((posconv_fwd_kernel<float, 48>), grid, dim3(512), 0, s, P);
        else hipLaunchKernelGGL((posconv_fwd_kernel<float, 64>), grid, dim3(512), 0, s, P);
    }
    return a8_check_launch("a8_posconv_fwd");
}

extern "C" int a8_posconv_wgrad(int dtype, const void* x, const void* dz, float* dw, int B, int T_, int d, int groups, int k,
                                int pad_left, void* stream) {
    A8_REQUIRE(k == KW, "a8_posconv_wgrad: kernel width must be %d (got %d)", KW, k);
    A8_REQUIRE(groups >= 1 && d % groups == 0, "a8_posconv_wgrad: d=%d not divisible by groups=%d", d, groups);
    const int dg = d / groups;
    A8_REQUIRE(dg == 48 || dg == 64, "a8_posconv_wgrad: channels per group must be 48 or 64 (got %d)", dg);
    A8_REQUIRE(dtype == A8_F32 || dtype == A8_BF16, "a8_posconv_wgrad: bad dtype");
    const dim3 grid(KW / WG_TAPS, groups, g_a8_tuning[6] > 0 ? g_a8_tuning[6] : 1);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == A8_BF16) {
        if (dg == 48) hipLaunchKernelGGL((posconv_wgrad_kernel<bf16, 48>), grid, dim3(256), 0, s, (const char*)x, (const char*)dz, dw, B, T_, d, pad_left);
        else hipLaunchKernelGGL((posconv_wgrad_kernel<bf16, 64>), grid, dim3(256), 0, s, (const char*)x, (const char*)dz, dw, B, T_, d, pad_left);
    } else {
        if (dg == 48) hipLaunchKernelGGL((posconv_wgrad_kernel<float, 48>), grid, dim3(256), 0, s, (const char*)x, (const char*)dz, dw, B, T_, d, pad_left);
        else hipLaunchKernelGGL((posconv_wgrad_kernel<float, 64>), grid, dim3(256), 0, s, (const char*)x, (const char*)dz, dw, B, T_, d, pad_left);
    }
    return a8_check_launch("a8_posconv_wgrad");
}

extern "C" int a8_weightnorm_pack(int dtype, const float* v, const float* g, void* wf, void* wt, float* inv_norm, float* scratch,
                                  int d, int groups, int k, void* stream) {
    A8_REQUIRE(groups >= 1 && d % groups == 0, "a8_weightnorm_pack: d=%d not divisible by groups=%d", d, groups);
    const int dg = d / groups;
    A8_REQUIRE(k >= 16 && k <= WN_MAXK && (k & (k - 1)) == 0 && dg <= WN_MAXDG, "a8_weightnorm_pack: k=%d must be a power of two in 16..%d, d/groups=%d <= %d",
               k, WN_MAXK, dg, WN_MAXDG);
    A8_REQUIRE(dtype == A8_F32 || dtype == A8_BF16, "a8_weightnorm_pack: bad dtype");
    hipStream_t s = (hipStream_t)stream;
#define WN_PART(DG, KK) hipLaunchKernelGGL((wn_partial_kernel<DG, KK>), dim3(d), dim3(256), 0, s, v, (const float*)nullptr, scratch, dg, k)
    WN_DISPATCH(dg, k, WN_PART);
#undef WN_PART
    hipLaunchKernelGGL(wn_reduce_kernel, dim3(k / 16), dim3(1024), 0, s, (const float*)scratch, (const float*)nullptr, inv_norm, (float*)nullptr, d, k);
#define WN_PACK(DG, KK) do { \
        if (dtype == A8_BF16) hipLaunchKernelGGL((wn_pack_kernel<bf16, DG, KK>), dim3(d, 2), dim3(256), 0, s, v, g, (const float*)inv_norm, (bf16*)wf, (bf16*)wt, dg, k); \
        else hipLaunchKernelGGL((wn_pack_kernel<float, DG, KK>), dim3(d, 2), dim3(256), 0, s, v, g, (const float*)inv_norm, (float*)wf, (float*)wt, dg, k); } while (0)
    WN_DISPATCH(dg, k, WN_PACK);
#undef WN_PACK
    return a8_check_launch("a8_weightnorm_pack");
}

extern "C" int a8_weightnorm_bwd(const float* dwg, const float* v, const float* g, const float* inv_norm, float* dv, float* dgain,
                                 float* scratch, int d, int groups, int k, void* stream) {
    A8_REQUIRE(groups >= 1 && d % groups == 0, "a8_weightnorm_bwd: d=%d not divisible by groups=%d", d, groups);
    const int dg = d / groups;
    A8_REQUIRE(k >= 16 && k <= WN_MAXK && (k & (k - 1)) == 0 && dg <= WN_MAXDG, "a8_weightnorm_bwd: k=%d must be a power of two in 16..%d, d/groups=%d <= %d",
               k, WN_MAXK, dg, WN_MAXDG);
    hipStream_t s = (hipStream_t)stream;
    float* dot = scratch + (size_t)d * k;
#define WN_PART(DG, KK) hipLaunchKernelGGL((wn_partial_kernel<DG, KK>), dim3(d), dim3(256), 0, s, v, dwg, scratch, dg, k)
    WN_DISPATCH(dg, k, WN_PART);
#undef WN_PART
    hipLaunchKernelGGL(wn_reduce_kernel, dim3(k / 16), dim3(1024), 0, s, (const float*)scratch, inv_norm, dot, dgain, d, k);
#define WN_BWD(DG, KK) hipLaunchKernelGGL((wn_bwd_kernel<DG, KK>), dim3(d), dim3(256), 0, s, dwg, v, g, inv_norm, (const float*)dot, dv, dg, k)
    WN_DISPATCH(dg, k, WN_BWD);
#undef WN_BWD
    return a8_check_launch("a8_weightnorm_bwd");
}
