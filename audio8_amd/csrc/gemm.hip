// a8hip GEMM: C[M,N] = epi(alpha * A[M,K] . B[N,K]^T + bias) on MFMA, for bf16 and exact f32.
//
// One kernel template covers the three products the path needs (SURVEY 8a rows a1,a2,a5,a6,a7):
//   NT  A stored [M][K], B stored [N][K]      forward  y = x W^T        (Dense, conv-as-GEMM)
//   NN  A stored [M][K], B stored [K][N]      dgrad    dx = dy W
//   TN  A stored [K][M], B stored [K][N]      wgrad    dW = dy^T x      (split-K over the rows)
// An operand that is K-strided in memory ("KS") is staged [k][row] in LDS and read with
// ds_read_b64_tr_b16 (bf16) so no operand is ever transposed through HBM.
//
// Block = 256 threads = 4 waves (2x2), tile 128x128, BK = 64 (bf16) / 16 (f32), register-staged
// double-buffered LDS (global loads for tile t+1 are in flight while tile t is multiplied).
// "Row remap": an operand's outer index o maps to (o / rpb) * bs + (o % rpb) * ld, which lets the
// strided conv layers run as plain GEMMs over overlapping rows of a channels-last activation
// (row t of utterance b starts at b*Lin*C + t*stride*C and is k*C long).
#include "a8_common.h"
#include "a8hip_internal.h"

namespace {

struct Op { const char* p; long long ld, rpb, bs; };
struct GemmArgs {
    Op A, B;
    char* C; long long ldc, c_rpb, c_bs;
    const char* zin; char* zout; long long ldz;
    const float* bias;
    int M, N, K, kchunk, nsplit, tiles_m, tiles_n;
    int act, out_f32, atomic, bias_rows;
    float alpha;
};

A8_DEV long long outer_off(const Op& o, long long idx) {
    return o.rpb ? (idx / o.rpb) * o.bs + (idx % o.rpb) * o.ld : idx * o.ld;
}

template <typename T, bool KS> struct Stage {
    static constexpr int EPC = 16 / (int)sizeof(T);                  // elements per 16-byte chunk
    static constexpr int CPR = KS ? 128 / EPC : Tile<T>::BK / EPC;   // chunks per LDS row
    static constexpr int NCH = KS ? Tile<T>::BK * CPR / 256 : 128 * CPR / 256;
    static constexpr int RSTEP = 256 / CPR;                          // LDS rows between a thread's chunks
};

template <typename T, bool AKS, bool BKS>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs g) {
    constexpr int BK = Tile<T>::BK;
    constexpr int TILE_BYTES = sizeof(T) == 2 ? 16384 : 10240;
    typedef Mma<T> MM;
    typedef Stage<T, AKS> SA;
    typedef Stage<T, BKS> SB;
    __shared__ __attribute__((aligned(16))) char lds[4 * TILE_BYTES];

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int nblk = g.tiles_m * g.tiles_n * g.nsplit;
    const int bid = xcd_remap(blockIdx.x, nblk);
    const int split = bid / (g.tiles_m * g.tiles_n);
    const int tile = bid - split * (g.tiles_m * g.tiles_n);
    const int m0 = (tile / g.tiles_n) * 128, n0 = (tile % g.tiles_n) * 128;
    const int kbeg = split * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int nkb = (kend - kbeg + BK - 1) / BK;

    // ---- per-thread staging coordinates
    const int a_c = t % SA::CPR, a_r = t / SA::CPR;      // chunk within LDS row, first LDS row
    const int b_c = t % SB::CPR, b_r = t / SB::CPR;
    long long a_base[SA::NCH], b_base[SB::NCH];          // KC: element offset of the row start (-1 = out of range)
    if constexpr (!AKS) {
#pragma unroll
        for (int j = 0; j < SA::NCH; ++j) {
            const int row = m0 + a_r + j * SA::RSTEP;
            a_base[j] = row < g.M ? outer_off(g.A, row) : -1;
        }
    }
    if constexpr (!BKS) {
#pragma unroll
        for (int j = 0; j < SB::NCH; ++j) {
            const int row = n0 + b_r + j * SB::RSTEP;
            b_base[j] = row < g.N ? outer_off(g.B, row) : -1;
        }
    }
    u32x4 ra[SA::NCH], rb[SB::NCH];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    auto gload = [&](int kb) {
        const int k0 = kbeg + kb * BK;
#pragma unroll
        for (int j = 0; j < SA::NCH; ++j) {
            if constexpr (!AKS) {
                const int k = k0 + a_c * SA::EPC;
                ra[j] = (a_base[j] >= 0 && k + SA::EPC <= kend) ? ld16(g.A.p + (a_base[j] + k) * (long long)sizeof(T)) : zero4;
            } else {
                const int k = k0 + a_r + j * SA::RSTEP, m = m0 + a_c * SA::EPC;
                ra[j] = (k < kend && m + SA::EPC <= g.M) ? ld16(g.A.p + (outer_off(g.A, k) + m) * (long long)sizeof(T)) : zero4;
            }
        }
#pragma unroll
        for (int j = 0; j < SB::NCH; ++j) {
            if constexpr (!BKS) {
                const int k = k0 + b_c * SB::EPC;
                rb[j] = (b_base[j] >= 0 && k + SB::EPC <= kend) ? ld16(g.B.p + (b_base[j] + k) * (long long)sizeof(T)) : zero4;
            } else {
                const int k = k0 + b_r + j * SB::RSTEP, n = n0 + b_c * SB::EPC;
                rb[j] = (k < kend && n + SB::EPC <= g.N) ? ld16(g.B.p + (outer_off(g.B, k) + n) * (long long)sizeof(T)) : zero4;
            }
        }
    };
    auto lstore = [&](int buf) {
        char* ta = lds + buf * 2 * TILE_BYTES;
        char* tb = ta + TILE_BYTES;
#pragma unroll
        for (int j = 0; j < SA::NCH; ++j) {
            const int r = a_r + j * SA::RSTEP;
            st16(ta + (AKS ? Tile<T>::ks_off(r, a_c) : Tile<T>::kc_off(r, a_c)), ra[j]);
        }
#pragma unroll
        for (int j = 0; j < SB::NCH; ++j) {
            const int r = b_r + j * SB::RSTEP;
            st16(tb + (BKS ? Tile<T>::ks_off(r, b_c) : Tile<T>::kc_off(r, b_c)), rb[j]);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (nkb > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
        if (kb + 1 < nkb) gload(kb + 1);
        const char* ta = lds + buf * 2 * TILE_BYTES;
        const char* tb = ta + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < BK / MM::KI; ++ks) {
            typename MM::Frag fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = AKS ? Tile<T>::ks_frag(ta, wm * 64 + 16 * i, ks, lane) : Tile<T>::kc_frag(ta, wm * 64 + 16 * i, ks, lane);
                fb[i] = BKS ? Tile<T>::ks_frag(tb, wn * 64 + 16 * i, ks, lane) : Tile<T>::kc_frag(tb, wn * 64 + 16 * i, ks, lane);
            }
            // swapped operands: accumulator reg r <-> 4 consecutive n, lane&15 <-> m  (vector C stores)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = MM::mma(fb[j], fa[i], acc[i][j]);
        }
        if (kb + 1 < nkb) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue
    const int g4 = (lane >> 4) * 4, li = lane & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + 16 * i + li;
        if (m >= g.M) continue;
        const long long crow = g.c_rpb ? (m / g.c_rpb) * g.c_bs + (m % g.c_rpb) * g.ldc : (long long)m * g.ldc;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + 16 * j + g4;
            if (n >= g.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * g.alpha;
            if (g.bias) {
                if (g.bias_rows) {
                    const float bm = g.bias[m];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += bm;
                } else {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(g.bias + n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += bv[r];
                }
            }
            if (g.zout) {
                T* zp = reinterpret_cast<T*>(g.zout) + (long long)m * g.ldz + n;
#pragma unroll
                for (int r = 0; r < 4; ++r) zp[r] = from_f<T>(v[r]);
            }
            if (g.act == A8_ACT_GELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_f(v[r]);
            } else if (g.act == A8_ACT_MUL_GELU_GRAD) {
                const T* zp = reinterpret_cast<const T*>(g.zin) + (long long)m * g.ldz + n;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= gelu_grad_f(to_f(zp[r]));
            }
            if (g.out_f32) {
                float* cp = reinterpret_cast<float*>(g.C) + crow + n;
                if (g.atomic) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) atomicAdd(cp + r, v[r]);
                } else {
                    *reinterpret_cast<f32x4*>(cp) = (f32x4){v[0], v[1], v[2], v[3]};
                }
            } else {
                T* cp = reinterpret_cast<T*>(g.C) + crow + n;
                if constexpr (sizeof(T) == 2) {
                    bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                    *reinterpret_cast<bf16x4*>(cp) = o;
                } else {
                    *reinterpret_cast<f32x4*>(cp) = (f32x4){v[0], v[1], v[2], v[3]};
                }
            }
        }
    }
}

template <typename T> int launch(const a8_gemm_desc* d, const GemmArgs& g, int grid, hipStream_t s) {
    switch (d->layout) {
        case A8_GEMM_NT: hipLaunchKernelGGL((gemm_kernel<T, false, false>), dim3(grid), dim3(256), 0, s, g); break;
        case A8_GEMM_NN: hipLaunchKernelGGL((gemm_kernel<T, false, true>), dim3(grid), dim3(256), 0, s, g); break;
        case A8_GEMM_TN: hipLaunchKernelGGL((gemm_kernel<T, true, true>), dim3(grid), dim3(256), 0, s, g); break;
        default: return a8_set_error(A8_ERR_ARG, "a8_gemm: bad layout %d", d->layout);
    }
    return a8_check_launch("a8_gemm");
}

}  // namespace

extern "C" int a8_gemm(const a8_gemm_desc* d, void* stream) {
    A8_REQUIRE(d != nullptr, "a8_gemm: null descriptor");
    A8_REQUIRE(d->dtype == A8_F32 || d->dtype == A8_BF16, "a8_gemm: bad dtype %d", d->dtype);
    A8_REQUIRE(d->M >= 0 && d->N > 0 && d->K >= 0, "a8_gemm: bad sizes M=%d N=%d K=%d", d->M, d->N, d->K);
    if (d->M == 0) return A8_OK;
    const int esz = d->dtype == A8_BF16 ? 2 : 4, epc = 16 / esz;
    const bool aks = d->layout == A8_GEMM_TN, bks = d->layout != A8_GEMM_NT;
    A8_REQUIRE(d->N % 4 == 0, "a8_gemm: N=%d must be a multiple of 4", d->N);
    if (!aks || !bks) A8_REQUIRE(d->K % epc == 0, "a8_gemm: K=%d must be a multiple of %d for a K-contiguous operand", d->K, epc);
    if (aks) A8_REQUIRE(d->M % epc == 0, "a8_gemm: M=%d must be a multiple of %d (TN)", d->M, epc);
    if (bks) A8_REQUIRE(d->N % epc == 0, "a8_gemm: N=%d must be a multiple of %d (NN/TN)", d->N, epc);
    A8_REQUIRE(d->lda % epc == 0 && d->ldb % epc == 0 && d->ldc % 4 == 0, "a8_gemm: leading dimensions must keep 16-byte alignment");
    A8_REQUIRE(((uintptr_t)d->A % 16) == 0 && ((uintptr_t)d->B % 16) == 0 && ((uintptr_t)d->C % 16) == 0, "a8_gemm: pointers must be 16-byte aligned");
    A8_REQUIRE(d->a_bs % epc == 0 && d->b_bs % epc == 0, "a8_gemm: batch strides must keep 16-byte alignment");
    const int bk = d->dtype == A8_BF16 ? 64 : 16;
    GemmArgs g;
    g.A = Op{(const char*)d->A, d->lda, d->a_rpb, d->a_bs};
    g.B = Op{(const char*)d->B, d->ldb, d->b_rpb, d->b_bs};
    g.C = (char*)d->C; g.ldc = d->ldc; g.c_rpb = d->c_rpb; g.c_bs = d->c_bs;
    g.zin = (const char*)d->zin; g.zout = (char*)d->zout; g.ldz = d->ldz;
    g.bias = d->bias; g.bias_rows = d->bias_rows;
    g.M = d->M; g.N = d->N; g.K = d->K;
    g.tiles_m = cdiv(d->M, 128); g.tiles_n = cdiv(d->N, 128);
    int nsplit = d->splitk > 0 ? d->splitk : 1;
    int kchunk = cdiv(cdiv(d->K, nsplit), bk) * bk;
    if (kchunk == 0) kchunk = bk;
    nsplit = d->K > 0 ? cdiv(d->K, kchunk) : 1;
    g.kchunk = kchunk; g.nsplit = nsplit;
    g.act = d->act; g.out_f32 = d->out_f32; g.alpha = d->alpha;
    g.atomic = (nsplit > 1 || d->accumulate) ? 1 : 0;
    A8_REQUIRE(!(g.atomic && !d->out_f32), "a8_gemm: split-K / accumulate needs an f32 output");
    A8_REQUIRE(!(nsplit > 1 && (d->bias || d->act || d->zout)), "a8_gemm: split-K cannot carry a bias/activation epilogue");
    A8_REQUIRE(d->act != A8_ACT_MUL_GELU_GRAD || d->zin, "a8_gemm: MUL_GELU_GRAD needs zin");
    const long long grid = (long long)g.tiles_m * g.tiles_n * nsplit;
    A8_REQUIRE(grid < (1ll << 30), "a8_gemm: grid too large");
    hipStream_t s = (hipStream_t)stream;
    return d->dtype == A8_BF16 ? launch<bf16>(d, g, (int)grid, s) : launch<float>(d, g, (int)grid, s);
}
