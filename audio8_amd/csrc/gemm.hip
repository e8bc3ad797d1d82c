// a8hip GEMM: C[M,N] = epi(alpha * A[M,K] . B[N,K]^T + bias) on MFMA, for bf16 and exact f32.
//
// One kernel template covers the three products the path needs (SURVEY 8a rows a1,a2,a5,a6,a7):
//   NT  A stored [M][K], B stored [N][K]      forward  y = x W^T        (Dense, conv-as-GEMM)
//   NN  A stored [M][K], B stored [K][N]      dgrad    dx = dy W
//   TN  A stored [K][M], B stored [K][N]      wgrad    dW = dy^T x      (split-K over the rows)
// An operand that is K-strided in memory ("KS") is staged [k][row] in LDS and read with
// ds_read_b64_tr_b16 (bf16) so no operand is ever transposed through HBM.
//
// Block = 512 threads = 8 waves (4 along M x 2 along N, each 32x64 of the 128x128 tile; 4 waves per SIMD at two
// blocks per CU hide the LDS / DMA latencies).  A k-tile = 128 bytes of K per row (64 bf16), staged by LDS-DMA
// (global_load_lds, 16 B per lane) into a 2-deep LDS ring behind a counted s_waitcnt vmcnt and ONE raw s_barrier
// per k-tile; every fragment read of a k-tile is issued ahead of its MFMAs.  (f32: 64-byte rows, 4-deep ring, 4 waves.)
// a8_gemm_grouped runs several independent problems of one layout in ONE launch (the four weight-gradient GEMMs
// of an encoder layer: 432 tiles fill the chip without split-K, where each alone needs 3-12 way split-K + atomics).
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
    const char* zin; char* zout; long long ldz, z_rpb, z_bs;
    const float* bias;
    float* colsum;
    int M, N, K, kchunk, nsplit, tiles_m, tiles_n;
    int act, out_f32, atomic, bias_rows, group_m, direct_off;
#ifdef A8_STAMPS
    unsigned long long* stamps;                   // developer build only (make STAMPS=1, a8_debug_set_stamps): 8 timestamps per block, nullptr = off
#endif
    float alpha;
};


// (index / rows_per_batch, index % rows_per_batch) -> offset.  Indices and rows_per_batch fit 31 bits (checked on the host): one
// 32-bit unsigned division instead of the ~200-instruction 64-bit one the compiler emits for long long operands.
A8_DEV long long remap_off(long long idx, long long rpb, long long bs, long long ld) {
    if (!rpb) return idx * ld;
    const unsigned q = (unsigned)idx / (unsigned)rpb, r = (unsigned)idx - q * (unsigned)rpb;
    return (long long)q * bs + (long long)r * ld;
}
A8_DEV long long outer_off(const Op& o, long long idx) { return remap_off(idx, o.rpb, o.bs, o.ld); }

// 16 zero bytes: the LDS-DMA source of every out-of-range lane (M/N/K tails are zero-filled this way)
__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

A8_DEV void glds16(const char* src, char* lds_dst_wave_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst_wave_uniform, 16, 0, 0);
}

// ---- stage tile geometry.  An operand stage is 128 rows x KROW bytes of K (KROW = 64 or 128), filled by LDS-DMA
// pieces of 1 KB (one wave-instruction each, lane l lands at piece + 16 l).  The XOR swizzles live on the per-lane
// SOURCE address; the LDS image is linear.
//   KC (reduction contiguous): [128 rows][KROW]; lane -> row RPP*p + l/SPR, slot l%SPR holds chunk slot ^ swz(row)
//   KS bf16: [BK k][256 B]; piece = 4 k-rows; lane -> k 4p + l/16, 16-B piece l%16; 32-B chunk c stored at c ^ key(k)
//   KS f32 : [BK k][512 B]; piece = 2 k-rows; lane -> k 2p + l/32, 16-B piece l%32 (no swizzle)
template <int KB> struct KcGeom {
    static constexpr int KROW = 64 * KB, SPR = KROW / 16, RPP = 1024 / KROW, OPBYTES = 128 * KROW, NPIECE = OPBYTES / 1024;
    static A8_DEV int swz(int row) { return KB == 1 ? ((row >> 1) & 3) : (row & 7); }     // (one 16-byte slot per row of a 16-row fragment)
};
template <typename T, int KB> struct StageTile;
template <int KB> struct StageTile<bf16, KB> {
    typedef KcGeom<KB> G;
    static constexpr int BK = 32 * KB, KSTEPS = KB;
    static A8_DEV int ks_key(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }
    static A8_DEV bf16x8 kc_frag(const char* t, int rb, int ks, int lane) {
        const int row = rb + (lane & 15);
        return *reinterpret_cast<const bf16x8*>(t + row * G::KROW + (((ks * 4 + (lane >> 4)) ^ G::swz(row)) << 4));
    }
    // The two transposed reads of a fragment as inline assembly: in front of the ds_read_tr builtin (no alias information) hipcc puts
    // s_waitcnt vmcnt(0), which drained the LDS-DMA ring at the top of EVERY k-tile of the K-strided layouts (weight gradients): the
    // tile requested last had to land at once and the refill could only be issued behind the reads.  The callers wait for these reads
    // themselves (s_waitcnt lgkmcnt in front of the MFMAs: the compiler does not track an asm statement's LDS result).
    static A8_DEV bf16x8 ks_frag(const char* t, int rb, int ks, int lane) {
        const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3, nb = rb >> 4;
        const int k0 = ks * 32 + 8 * g + q;            // the second read is k0 + 4: same swizzle key, 1024 bytes on
        const unsigned a = (unsigned)(size_t)LDS_PTR(char, t + k0 * 256 + ((nb ^ ks_key(k0)) << 5) + 8 * p);
        u32x2 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(hi) : "v"(a));
        const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
        return __builtin_bit_cast(bf16x8, v);
    }
    static A8_DEV void ks_src(int p, int lane, int& k, int& col) {
        k = 4 * p + (lane >> 4);                         // 16 lanes (16-byte pieces) per k-row
        const int pc = lane & 15;
        col = (((pc >> 1) ^ ks_key(k)) << 4) + ((pc & 1) << 3);
    }
};
template <int KB> struct StageTile<float, KB> {
    typedef KcGeom<KB> G;
    static constexpr int BK = 16 * KB, KSTEPS = 4 * KB;
    static A8_DEV float kc_frag(const char* t, int rb, int ks, int lane) {
        const int row = rb + (lane & 15);
        return *reinterpret_cast<const float*>(t + row * G::KROW + ((ks ^ G::swz(row)) << 4) + ((lane >> 4) << 2));
    }
    static A8_DEV float ks_frag(const char* t, int rb, int ks, int lane) {
        return *reinterpret_cast<const float*>(t + (ks * 4 + (lane >> 4)) * 512 + (rb + (lane & 15)) * 4);
    }
    static A8_DEV void ks_src(int p, int lane, int& k, int& col) {
        k = 2 * p + (lane >> 5);
        col = (lane & 31) << 2;
    }
};
// ---- epilogue (shared by the main-loop variants).  Needs NWM*2 waves x (16*MI/NP rows x 64 cols x 4 B) of free LDS;
// NP = passes over the wave's rows (2 halves the staging so that small-LDS variants keep their occupancy).
template <typename T> struct Vec8 { typedef T type __attribute__((ext_vector_type(8))); };

// zpre (16-bit outputs, NP == 1, when use_zpre): the lane's zin values, loaded by the caller ahead of its main loop
constexpr int NJ = 4;              // 16-column fragments per wave along N: a wave owns (16 MI) x 64 of the 128 x 128 tile

template <typename T, int MI, int NWM, int NP = 1>
A8_DEV void gemm_epilogue(const GemmArgs& g, f32x4 (&acc)[MI][NJ], char* lds, const int w, const int m0, const int n0, const int lane,
                          const typename Vec8<T>::type (&zpre)[2 * MI], const bool use_zpre, const float (&bpre)[8], const bool use_bpre) {
    const int wm = w >> 1, wn = w & 1;
    // ---- epilogue through LDS: each wave parks its 64-column f32 sub-tile in its own slab of the (now free) ring, then
    // walks it row by row so that every global store / f32 atomic wave-instruction is ONE contiguous row segment
    // (256 B of f32, 128 B of bf16) instead of 16 rows x 16 B -- the shape the memory-side atomic units need.
    constexpr int WROWS = 16 * MI;                                    // rows of the tile owned by a wave
    constexpr int MP = MI / NP, PROWS = 16 * MP;                      // fragments / rows per pass
    static_assert(MI % NP == 0, "passes must divide the wave's row fragments");
    float* cw = reinterpret_cast<float*>(lds) + w * (PROWS * 64);
    constexpr int WN = 16 * NJ;                                       // columns of the tile owned by a wave (its LDS slab stays 64 wide)
    const int nw = n0 + wn * WN;
    auto zrow = [&](int m) -> long long { return remap_off(m, g.z_rpb, g.z_bs, g.ldz); };
    const bool f32_rows = g.out_f32 || sizeof(T) == 4;
    float csum1 = 0.f;
    float csum8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
        if (ps) __builtin_amdgcn_wave_barrier();
        {
            const int g4 = lane >> 4, li = lane & 15;
#pragma unroll
            for (int i = 0; i < MP; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    f32x4 v = acc[ps * MP + i][j];
                    v *= g.alpha;
                    *reinterpret_cast<f32x4*>(cw + (16 * i + li) * 64 + (((4 * j + g4) ^ li) << 2)) = v;
                }
        }
        __builtin_amdgcn_wave_barrier();      // the tile is wave-private: program order of this wave's LDS ops suffices
        const int mw = m0 + wm * WROWS + ps * PROWS;
        if (f32_rows) {
            const int n = nw + lane;
            const bool n_ok = n < g.N && lane < WN;
            const float bn = (g.bias && !g.bias_rows && n_ok) ? g.bias[n] : 0.f;
            for (int rr = 0; rr < PROWS; ++rr) {
                const int m = mw + rr;
                if (m >= g.M) break;
                float v = cw[rr * 64 + ((((lane >> 2) ^ (rr & 15))) << 2) + (lane & 3)] + bn;
                if (g.bias && g.bias_rows) v += g.bias[m];
                if (!n_ok) continue;
                if (g.zout) reinterpret_cast<T*>(g.zout)[zrow(m) + n] = from_f<T>(v);
                if (g.act == A8_ACT_GELU) v = gelu_f(v);
                else if (g.act == A8_ACT_MUL_GELU_GRAD) v *= gelu_grad_f(to_f(reinterpret_cast<const T*>(g.zin)[zrow(m) + n]));
                else if (g.act == A8_ACT_ADD) v += to_f(reinterpret_cast<const T*>(g.zin)[zrow(m) + n]);
                const long long crow = remap_off(m, g.c_rpb, g.c_bs, g.ldc);
                float* cp = reinterpret_cast<float*>(g.C) + crow + n;
                if (g.atomic) atomicAdd(cp, v);
                else *cp = v;
                csum1 += v;
            }
        } else {
            // 16-bit output: lane owns 8 consecutive columns (16 B), a wave-instruction covers 8 rows x 128 B
            const int c8 = (lane & 7) << 3;                 // first column within the wave's 64
            const int n = nw + c8;
            const bool n_ok = n < g.N && c8 < WN;
            float bn[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bn[e] = use_bpre ? bpre[e] : ((g.bias && !g.bias_rows && n_ok) ? g.bias[n + e] : 0.f);
            typedef T T8 __attribute__((ext_vector_type(8)));
#pragma unroll
            for (int it = 0; it < PROWS / 8; ++it) {
                if (mw + 8 * it >= g.M) break;
                const int rr = 8 * it + (lane >> 3);
                const int m = mw + rr;
                const int ch = c8 >> 2;                      // even 16-byte chunk index; its partner is ch + 1
                const f32x4 p0 = *reinterpret_cast<const f32x4*>(cw + rr * 64 + ((ch ^ (rr & 15)) << 2));
                const f32x4 p1 = *reinterpret_cast<const f32x4*>(cw + rr * 64 + (((ch + 1) ^ (rr & 15)) << 2));
                if (m >= g.M || !n_ok) continue;
                float v[8] = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
                const float bm = (g.bias && g.bias_rows) ? g.bias[m] : 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += bn[e] + bm;
                if (g.zout) {
                    T8 zv;
#pragma unroll
                    for (int e = 0; e < 8; ++e) zv[e] = from_f<T>(v[e]);
                    *reinterpret_cast<T8*>(reinterpret_cast<T*>(g.zout) + zrow(m) + n) = zv;
                }
                if (g.act == A8_ACT_GELU) {
                    gelu_n(v);
                } else if (g.act == A8_ACT_MUL_GELU_GRAD) {
                    const T8 zz = use_zpre ? zpre[NP == 1 ? it : 0] : *reinterpret_cast<const T8*>(reinterpret_cast<const T*>(g.zin) + zrow(m) + n);
                    float zf[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) zf[e] = to_f(zz[e]);
                    mul_gelu_grad_n(v, zf);
                } else if (g.act == A8_ACT_ADD) {
                    const T8 zz = use_zpre ? zpre[NP == 1 ? it : 0] : *reinterpret_cast<const T8*>(reinterpret_cast<const T*>(g.zin) + zrow(m) + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += to_f(zz[e]);
                }
                const long long crow = remap_off(m, g.c_rpb, g.c_bs, g.ldc);
                T8 ov;
#pragma unroll
                for (int e = 0; e < 8; ++e) { ov[e] = from_f<T>(v[e]); csum8[e] += v[e]; }
                *reinterpret_cast<T8*>(reinterpret_cast<T*>(g.C) + crow + n) = ov;
            }
        }
    }
    if (g.colsum) {                                      // block-uniform: combine the NWM row-waves in LDS, one atomic per column per block
        if (f32_rows) {
            cw[lane] = csum1;                            // the wave is done with its staging area
        } else {                                         // lanes l, l+8, .. l+56 hold the same 8 columns
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = csum8[e];
                t += __shfl_xor(t, 8, 64);
                t += __shfl_xor(t, 16, 64);
                t += __shfl_xor(t, 32, 64);
                csum8[e] = t;
            }
            if (lane < 8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) cw[lane * 8 + e] = csum8[e];
            }
        }
        __syncthreads();
        if (wm == 0) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < NWM; ++q) t += reinterpret_cast<const float*>(lds)[(q * 2 + wn) * (PROWS * 64) + lane];
            if (nw + lane < g.N && lane < WN) atomicAdd(g.colsum + nw + lane, t);
        }
    }
}
// ---- epilogue without LDS (16-bit outputs of the NT kernels; the arithmetic, in the order of gemm_epilogue: bit-identical results).
// Accumulator register r of fragment (i, j) is column 16 j + 4 (lane >> 4) + r of row 16 i + (lane & 15): two v_permlane16_swap per
// fragment PAIR give every lane 8 consecutive columns, so a wave-instruction stores 16 rows x 64 contiguous bytes (gemm_stream.hip's
// epilogue).  No barrier, no wait for the other waves, nothing parked in the ring: a wave stores as soon as its own MFMAs are done.
// epre: the lane's prefetched second input -- zin in the layout of the stores (epre[2 i + jj]; the same two swaps, an involution, turn
// it into the accumulator layout) or the bias in the accumulator layout (epre[2 jj + q] = 4 floats); never both (the caller checks).
A8_DEV unsigned pack_bf16x2(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 v = {(bf16)a, (bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
template <int MI>
A8_DEV void gemm_epilogue_direct(const GemmArgs& g, f32x4 (&acc)[MI][NJ], const int w, const int m0, const int n0, const int lane,
                                 const u32x4 (&epre)[2 * MI]) {
#pragma clang fp contract(off)
    static_assert(MI == 2 && NJ == 4, "a wave owns 32 x 64");
    const int wm = w >> 1, wn = w & 1, li = lane & 15, gq = lane >> 4;
    const bool zmul = g.act == A8_ACT_MUL_GELU_GRAD, zadd = g.act == A8_ACT_ADD, gelu = g.act == A8_ACT_GELU;
    const bool has_z = zmul || zadd;
    bf16* const C = reinterpret_cast<bf16*>(g.C);
    bf16* const Z = reinterpret_cast<bf16*>(g.zout);
    long long crow[MI], zrow[MI];
    bool rok[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * (16 * MI) + 16 * i + li;
        rok[i] = m < g.M;
        crow[i] = remap_off(rok[i] ? m : 0, g.c_rpb, g.c_bs, g.ldc);
        zrow[i] = Z ? remap_off(rok[i] ? m : 0, g.z_rpb, g.z_bs, g.ldz) : 0;
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int col = n0 + wn * 64 + 32 * jj + 16 * (gq & 1) + 8 * (gq >> 1);       // first of the lane's 8 stored columns
        const bool cok = col < g.N;
        float bz[2][4];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) bz[q][e] = (g.bias && !has_z) ? __uint_as_float(epre[2 * jj + q][e]) : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            unsigned zl[2] = {0u, 0u}, zh[2] = {0u, 0u}, yl[2], yh[2];
            if (has_z) {
                const u32x4 zv = epre[2 * i + jj];
                auto t0 = __builtin_amdgcn_permlane16_swap(zv[0], zv[2], false, false);
                auto t1 = __builtin_amdgcn_permlane16_swap(zv[1], zv[3], false, false);
                zl[0] = t0[0]; zl[1] = t0[1]; zh[0] = t1[0]; zh[1] = t1[1];
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float x[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { x[e] = acc[i][2 * jj + q][e] * g.alpha; x[e] += bz[q][e] + 0.f; }
                if (has_z) {
                    const float zf[4] = {__uint_as_float(zl[q] << 16), __uint_as_float(zl[q] & 0xFFFF0000u),
                                         __uint_as_float(zh[q] << 16), __uint_as_float(zh[q] & 0xFFFF0000u)};
                    if (zmul) mul_gelu_grad_n(x, zf);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[e] += zf[e];
                    }
                } else if (Z) { zl[q] = pack_bf16x2(x[0], x[1]); zh[q] = pack_bf16x2(x[2], x[3]); }
                if (gelu) gelu_n(x);
                yl[q] = pack_bf16x2(x[0], x[1]); yh[q] = pack_bf16x2(x[2], x[3]);
            }
            {
                auto s0 = __builtin_amdgcn_permlane16_swap(yl[0], yl[1], false, false);
                auto s1 = __builtin_amdgcn_permlane16_swap(yh[0], yh[1], false, false);
                const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                if (rok[i] && cok) *reinterpret_cast<u32x4*>(C + crow[i] + col) = o;
            }
            if (Z && !has_z) {
                auto s0 = __builtin_amdgcn_permlane16_swap(zl[0], zl[1], false, false);
                auto s1 = __builtin_amdgcn_permlane16_swap(zh[0], zh[1], false, false);
                const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                if (rok[i] && cok) *reinterpret_cast<u32x4*>(Z + zrow[i] + col) = o;
            }
        }
    }
}

// Main loop: NST-deep ring of LDS stages filled by LDS-DMA (global_load_lds, 16 B per lane), NST-1 k-tiles in
// flight per block; one raw s_barrier and one counted s_waitcnt vmcnt per k-tile; no register staging.
#ifdef A8_STAMPS
A8_DEV void stamp(const GemmArgs& g, int i) {
    if (g.stamps && threadIdx.x == 0) {
        g.stamps[(long long)blockIdx.x * 8 + i] = __builtin_amdgcn_s_memrealtime();
        if (i == 2) g.stamps[(long long)blockIdx.x * 8 + 5] = __builtin_readcyclecounter();      // shader-clock cycles over the main loop:
        if (i == 3) g.stamps[(long long)blockIdx.x * 8 + 6] = __builtin_readcyclecounter();      // with the 100 MHz stamps, the clock it ran at
    }
}
#else
A8_DEV void stamp(const GemmArgs&, int) {}
#endif

// A3: the asymmetric ring (A three k-tiles deep, B two) of gemm_a3_kernel below
// DIRECT: 16-bit NT outputs leave through gemm_epilogue_direct (no LDS, no barrier) whenever the problem allows (no f32 / atomic / column-sum
// / per-row-bias output, not bias AND second input together); a8_set_tuning key 22 = 1 restores the LDS-staged epilogue (A/B runs)
// KFAST (NT, asymmetric ring; the host launches it when every k range is whole k-tiles): a piece's source is a fixed per-lane pointer plus a
// block-uniform k offset -- one 64-bit add per piece per k-tile instead of a product, two compares and two selects; M / N tails by
// clamped rows (valid bytes; the surplus rows / columns of a tile are never stored or summed) instead of the zero-fill predicate.
template <typename T, bool AKS, bool BKS, int KB, int NST, int NWM, bool A3 = false, bool DIRECT = false, bool KFAST = false>
A8_DEV void gemm_core(const GemmArgs& g, const int tile, const int kbeg, const int kend) {
    typedef StageTile<T, KB> ST;
    typedef KcGeom<KB> G;
    typedef Mma<T> MM;
    constexpr int BK = ST::BK, EPC = 16 / (int)sizeof(T), SZ = (int)sizeof(T);
    constexpr int NW = 2 * NWM;                   // waves per block: NWM along M x 2 along N
    constexpr int MI = NWM == 2 ? 4 : 2;          // 16-row fragments per wave along M: a wave owns (16 MI) x 64 of the tile
    constexpr int BM = 16 * MI * NWM, BN = 32 * NJ;
    static_assert(BM == 128 && BN == 128, "128 x 128 tiles");
    constexpr int TA = BM * G::KROW, TB = BN * G::KROW, STAGE = TA + TB;
    constexpr int PPA = TA / 1024 / NW, PPB = TB / 1024 / NW;              // 1 KB LDS-DMA pieces per operand per stage per wave
    static_assert(PPA * NW * 1024 == TA && PPB * NW * 1024 == TB && PPA == PPB, "the waves share the DMA pieces evenly");
    constexpr int PPW = PPA;
    static_assert(!A3 || NST == 2, "the 3-deep A ring goes with a 2-deep B ring");
    constexpr int RING = A3 ? 3 * TA + 2 * TB : NST * STAGE;
    constexpr int NP = (KB == 1 && NST == 2) ? 2 : 1;                      // small-LDS variant: epilogue in two passes
    constexpr int EPI = NW * 16 * MI * 64 * 4 / NP;                        // epilogue staging: a 64-column f32 slab per wave
    __shared__ __attribute__((aligned(1024))) char lds[RING > EPI ? RING : EPI];

    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = w >> 1, wn = w & 1;            // wm in [0, NWM)
    stamp(g, 0);
    // grouped tile order: consecutive ids (= one XCD's co-resident blocks) cover GROUP_M m-tiles x several n-tiles, so
    // both the A row-panels and the B tiles they touch stay resident in that XCD's 4 MB L2 (sweeping all n-tiles per
    // m-panel streamed the whole weight matrix through L2 once per panel: 5-10x over-fetch measured with FETCH_SIZE).
    const int GROUP_M = g.group_m;
    const int tpg = GROUP_M * g.tiles_n;
    const int grp = tile / tpg, first_m = grp * GROUP_M;
    const int gsz = min(GROUP_M, g.tiles_m - first_m);
    const int tin = tile - grp * tpg;
    const int m0 = (first_m + tin % gsz) * BM, n0 = (tin / gsz) * BN;
    const int nkb = (kend - kbeg + BK - 1) / BK;
    const char* zero = reinterpret_cast<const char*>(g_zero16);

    // ---- per-lane LDS-DMA sources for this wave's pieces of each operand
    const char* a_ptr[PPW]; int a_k[PPW]; bool a_ok[PPW];
    const char* b_ptr[PPW]; int b_k[PPW]; bool b_ok[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int p = w + NW * j;
        if constexpr (!AKS) {
            const int rl = G::RPP * p + lane / G::SPR, row = m0 + rl;
            a_k[j] = (((lane % G::SPR) ^ G::swz(rl)) & (G::SPR - 1)) * EPC;
            a_ok[j] = row < g.M;
            a_ptr[j] = g.A.p + (a_ok[j] ? outer_off(g.A, row) : 0) * SZ;
        } else {
            int kl, col;
            ST::ks_src(p, lane, kl, col);
            a_k[j] = kl;
            a_ok[j] = m0 + col + EPC <= g.M;
            a_ptr[j] = g.A.p + (long long)(m0 + col) * SZ;
        }
        if constexpr (!BKS) {
            const int rl = G::RPP * p + lane / G::SPR, row = n0 + rl;
            b_k[j] = (((lane % G::SPR) ^ G::swz(rl)) & (G::SPR - 1)) * EPC;
            b_ok[j] = row < g.N;
            b_ptr[j] = g.B.p + (b_ok[j] ? outer_off(g.B, row) : 0) * SZ;
        } else {
            int kl, col;
            ST::ks_src(p, lane, kl, col);
            b_k[j] = kl;
            b_ok[j] = n0 + col + EPC <= g.N;
            b_ptr[j] = g.B.p + (long long)(n0 + col) * SZ;
        }
    }

    // K-strided operands walk their outer index by BK per k-tile: each lane keeps a running source pointer (one
    // 64-bit add per DMA piece per k-tile); a batch remap (rows_per_batch != 0, the conv weight gradient) only adds
    // a 32-bit row counter and a wrap correction.  a_k/b_k then hold the lane's CURRENT k.
    const char* a_cur[PPW]; int a_r[PPW];
    const char* b_cur[PPW]; int b_r[PPW];
    const long long a_step = (long long)BK * g.A.ld * SZ, a_wrap = (g.A.bs - g.A.rpb * g.A.ld) * SZ;
    const long long b_step = (long long)BK * g.B.ld * SZ, b_wrap = (g.B.bs - g.B.rpb * g.B.ld) * SZ;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        if constexpr (AKS) {
            const long long k = kbeg + a_k[j];
            a_cur[j] = a_ptr[j] + outer_off(g.A, k) * SZ;
            a_r[j] = g.A.rpb ? (int)((unsigned)k % (unsigned)g.A.rpb) : 0;
            a_k[j] = (int)k;
        }
        if constexpr (BKS) {
            const long long k = kbeg + b_k[j];
            b_cur[j] = b_ptr[j] + outer_off(g.B, k) * SZ;
            b_r[j] = g.B.rpb ? (int)((unsigned)k % (unsigned)g.B.rpb) : 0;
            b_k[j] = (int)k;
        }
    }
    auto issue = [&](int kb) {
        char* ta = lds + (kb % NST) * STAGE;
        char* tb = ta + TA;
        const int k0 = kbeg + kb * BK;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int p = w + NW * j;
            const char* sa;
            if constexpr (!AKS) {
                const int k = k0 + a_k[j];
                sa = (a_ok[j] && k + EPC <= kend) ? a_ptr[j] + (long long)k * SZ : zero;
            } else {
                sa = (a_ok[j] && a_k[j] < kend) ? a_cur[j] : zero;
                a_k[j] += BK;
                a_cur[j] += a_step;
                if (g.A.rpb) { a_r[j] += BK; while (a_r[j] >= (int)g.A.rpb) { a_r[j] -= (int)g.A.rpb; a_cur[j] += a_wrap; } }
            }
            glds16(sa, ta + p * 1024);
            const char* sb;
            if constexpr (!BKS) {
                const int k = k0 + b_k[j];
                sb = (b_ok[j] && k + EPC <= kend) ? b_ptr[j] + (long long)k * SZ : zero;
            } else {
                sb = (b_ok[j] && b_k[j] < kend) ? b_cur[j] : zero;
                b_k[j] += BK;
                b_cur[j] += b_step;
                if (g.B.rpb) { b_r[j] += BK; while (b_r[j] >= (int)g.B.rpb) { b_r[j] -= (int)g.B.rpb; b_cur[j] += b_wrap; } }
            }
            glds16(sb, tb + p * 1024);
        }
    };

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // The epilogues that read a second input (x gelu'(z), + residual) get it EARLY: right behind the first k-tile's DMAs (whose
    // issue they must not delay), so the loop's first counted wait retires them together with that k-tile (their latency would otherwise sit, exposed,
    // between the last MFMA and the stores of every tile).  16-bit rows only; the lane -> (row, 8 columns) map of gemm_epilogue.
    typedef typename Vec8<T>::type T8z;
    constexpr bool ZPRE = sizeof(T) == 2 && NP == 1;
    T8z zpre[2 * MI];
    bool use_zpre = false;
    float bpre[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // the lane's 8 bias columns, for the same reason
    bool use_bpre = false;
    constexpr bool DIRECT_OK = DIRECT && ZPRE && !AKS && !BKS && MI == 2;
    bool direct = false;                                             // block-uniform (kernel arguments only)
    if constexpr (DIRECT_OK) {
        const bool zact = g.act == A8_ACT_MUL_GELU_GRAD || g.act == A8_ACT_ADD;
        direct = !g.direct_off && !g.out_f32 && !g.atomic && !g.colsum && !g.bias_rows && !(g.bias && zact) && !(zact && g.zout);
    }
    auto prefetch_epilogue_inputs = [&]() {
        if constexpr (DIRECT_OK) {
            if (direct) {               // the direct epilogue's inputs, in ITS layouts, in the registers of zpre (never both kinds)
                const int li = lane & 15, gq = lane >> 4;
                if (g.act == A8_ACT_MUL_GELU_GRAD || g.act == A8_ACT_ADD) {
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int m = m0 + wm * (16 * MI) + 16 * i + li, n = n0 + wn * 64 + 32 * jj + 16 * (gq & 1) + 8 * (gq >> 1);
                            zpre[2 * i + jj] = T8z{};
                            if (m < g.M && n < g.N) {
                                const long long zr = remap_off(m, g.z_rpb, g.z_bs, g.ldz);
                                zpre[2 * i + jj] = *reinterpret_cast<const T8z*>(reinterpret_cast<const T*>(g.zin) + zr + n);
                            }
                        }
                } else if (g.bias) {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int n = n0 + wn * 64 + 32 * jj + 16 * q + 4 * gq;
                            f32x4 b = {0.f, 0.f, 0.f, 0.f};
                            if (n < g.N) b = *reinterpret_cast<const f32x4*>(g.bias + n);
                            zpre[2 * jj + q] = __builtin_bit_cast(T8z, b);
                        }
                }
                return;
            }
        }
        if constexpr (ZPRE) {
            use_zpre = g.zin != nullptr && !g.out_f32 && (g.act == A8_ACT_MUL_GELU_GRAD || g.act == A8_ACT_ADD);
            if (use_zpre) {
                const int c8 = (lane & 7) << 3, n = n0 + wn * (16 * NJ) + c8;
#pragma unroll
                for (int it = 0; it < 2 * MI; ++it) {
                    const int m = m0 + wm * (16 * MI) + 8 * it + (lane >> 3);
                    zpre[it] = T8z{};
                    if (m < g.M && n < g.N && c8 < 16 * NJ) {
                        const long long zr = remap_off(m, g.z_rpb, g.z_bs, g.ldz);
                        zpre[it] = *reinterpret_cast<const T8z*>(reinterpret_cast<const T*>(g.zin) + zr + n);
                    }
                }
            }
        }
        if constexpr (ZPRE && !BKS) {                                     // (a bias rides on forward products: NT)
            use_bpre = g.bias != nullptr && !g.bias_rows && !g.out_f32;
            if (use_bpre) {
                const int c8 = (lane & 7) << 3, n = n0 + wn * (16 * NJ) + c8;
                if (n < g.N && c8 < 16 * NJ) {
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(g.bias + n), b1 = *reinterpret_cast<const f32x4*>(g.bias + n + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bpre[e] = b0[e]; bpre[e + 4] = b1[e]; }
                }
            }
        }
    };

    if constexpr (A3) {
        // Asymmetric ring: the A operand (activations, streamed once from HBM / the Infinity Cache) is requested TWO k-tiles
        // ahead into a 3-deep ring, the B operand (weights, L2-resident) one k-tile ahead into a 2-deep ring: 3 x 16 + 2 x 16 = 80 KB,
        // still two blocks per CU.  Per iteration a wave issues its B pieces first and its A pieces last, so "all but the PPA
        // youngest" (vmcnt counts in issue order) is exactly: A(kb) and B(kb) have landed, A(kb+1) may still be in flight.
        static_assert(!KFAST || (!AKS && !BKS), "KFAST: K-contiguous operands");
        const char* a_src[PPA];
        const char* b_src[PPB];
        if constexpr (KFAST) {
#pragma unroll
            for (int j = 0; j < PPW; ++j) {
                const int rl = G::RPP * (w + NW * j) + lane / G::SPR;
                a_src[j] = g.A.p + (outer_off(g.A, min(m0 + rl, g.M - 1)) + kbeg + a_k[j]) * SZ;
                b_src[j] = g.B.p + (outer_off(g.B, min(n0 + rl, g.N - 1)) + kbeg + b_k[j]) * SZ;
            }
        }
        auto issue_a = [&](int kb, char* ta) {
            if constexpr (KFAST) {
                const long long koff = (long long)min(kb, nkb - 1) * (BK * SZ);      // (look-ahead past the last k-tile re-reads it: valid, unused)
#pragma unroll
                for (int j = 0; j < PPA; ++j) glds16(a_src[j] + koff, ta + (w + NW * j) * 1024);
                return;
            }
            const int k0 = kbeg + kb * BK;
#pragma unroll
            for (int j = 0; j < PPA; ++j) {
                const char* sa;
                if constexpr (!AKS) {
                    const int k = k0 + a_k[j];
                    sa = (a_ok[j] && k + EPC <= kend) ? a_ptr[j] + (long long)k * SZ : zero;
                } else {                                  // K-strided: the lane's running source pointer (see `issue` below)
                    sa = (a_ok[j] && a_k[j] < kend) ? a_cur[j] : zero;
                    a_k[j] += BK;
                    a_cur[j] += a_step;
                    if (g.A.rpb) { a_r[j] += BK; while (a_r[j] >= (int)g.A.rpb) { a_r[j] -= (int)g.A.rpb; a_cur[j] += a_wrap; } }
                }
                glds16(sa, ta + (w + NW * j) * 1024);
            }
        };
        auto issue_b = [&](int kb, char* tb) {
            if constexpr (KFAST) {
                const long long koff = (long long)min(kb, nkb - 1) * (BK * SZ);
#pragma unroll
                for (int j = 0; j < PPB; ++j) glds16(b_src[j] + koff, tb + (w + NW * j) * 1024);
                return;
            }
            const int k0 = kbeg + kb * BK;
#pragma unroll
            for (int j = 0; j < PPB; ++j) {
                const char* sb;
                if constexpr (!BKS) {
                    const int k = k0 + b_k[j];
                    sb = (b_ok[j] && k + EPC <= kend) ? b_ptr[j] + (long long)k * SZ : zero;
                } else {
                    sb = (b_ok[j] && b_k[j] < kend) ? b_cur[j] : zero;
                    b_k[j] += BK;
                    b_cur[j] += b_step;
                    if (g.B.rpb) { b_r[j] += BK; while (b_r[j] >= (int)g.B.rpb) { b_r[j] -= (int)g.B.rpb; b_cur[j] += b_wrap; } }
                }
                glds16(sb, tb + (w + NW * j) * 1024);
            }
        };
        char* const bbase = lds + 3 * TA;
        issue_a(0, lds);
        issue_b(0, bbase);
        prefetch_epilogue_inputs();
        issue_a(1, lds + TA);
        stamp(g, 1);
        int sa = 0, sb = 0;                           // ring slots of the k-tile being consumed
        for (int kb = 0; kb < nkb; ++kb) {
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PPA) : "memory");
            __builtin_amdgcn_s_barrier();
            if (kb == 0) stamp(g, 2);
            const int sa2 = sa == 0 ? 2 : sa - 1;     // (sa + 2) % 3: the slot read in the previous iteration
            issue_b(kb + 1, bbase + (sb ^ 1) * TB);   // (beyond kend: 16 zero bytes per lane -- keeps the counts uniform)
            issue_a(kb + 2, lds + sa2 * TA);
            __builtin_amdgcn_sched_barrier(0);
            const char* ta = lds + sa * TA;
            const char* tb = bbase + sb * TB;
            typename MM::Frag fa[ST::KSTEPS][MI], fb[ST::KSTEPS][NJ];
#pragma unroll
            for (int ks = 0; ks < ST::KSTEPS; ++ks) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    fa[ks][i] = AKS ? ST::ks_frag(ta, wm * (16 * MI) + 16 * i, ks, lane) : ST::kc_frag(ta, wm * (16 * MI) + 16 * i, ks, lane);
#pragma unroll
                for (int i = 0; i < NJ; ++i)
                    fb[ks][i] = BKS ? ST::ks_frag(tb, wn * (16 * NJ) + 16 * i, ks, lane) : ST::kc_frag(tb, wn * (16 * NJ) + 16 * i, ks, lane);
            }
            if constexpr ((AKS || BKS) && sizeof(T) == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (inline-asm transposed reads)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < ST::KSTEPS; ++ks)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = MM::mma(fb[ks][j], fa[ks][i], acc[i][j]);
            sa = sa == 2 ? 0 : sa + 1;
            sb ^= 1;
        }
    } else {
#pragma unroll
        for (int s = 0; s < NST - 1; ++s) issue(s);
        stamp(g, 1);
        prefetch_epilogue_inputs();
        for (int kb = 0; kb < nkb; ++kb) {
            // stage kb has landed once all but the (NST-2) youngest stages' DMAs of this wave are done (2*PPW per stage per lane)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NST - 2) * (PPA + PPB)) : "memory");
            __builtin_amdgcn_s_barrier();
            if (kb == 0) stamp(g, 2);
            // DMA issue order: the refill first, then the fragment reads (with the transposed reads as inline assembly -- StageTile::ks_frag --
            // that holds for the K-strided layouts too: no compiler-inserted vmcnt(0) in front of them any more)
            issue(kb + NST - 1);                  // refills the buffer every wave finished reading last iteration
            __builtin_amdgcn_sched_barrier(0);
            const char* ta = lds + (kb % NST) * STAGE;
            const char* tb = ta + TA;
            // all fragment reads of the k-tile are issued up front (one exposed LDS latency per tile, counted lgkmcnt waits
            // after that), pinned ahead of the MFMA block
            typename MM::Frag fa[ST::KSTEPS][MI], fb[ST::KSTEPS][NJ];
#pragma unroll
            for (int ks = 0; ks < ST::KSTEPS; ++ks) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    fa[ks][i] = AKS ? ST::ks_frag(ta, wm * (16 * MI) + 16 * i, ks, lane) : ST::kc_frag(ta, wm * (16 * MI) + 16 * i, ks, lane);
#pragma unroll
                for (int i = 0; i < NJ; ++i)
                    fb[ks][i] = BKS ? ST::ks_frag(tb, wn * (16 * NJ) + 16 * i, ks, lane) : ST::kc_frag(tb, wn * (16 * NJ) + 16 * i, ks, lane);
            }
            if constexpr ((AKS || BKS) && sizeof(T) == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (inline-asm transposed reads)
            __builtin_amdgcn_sched_barrier(0);
            // swapped operands: accumulator reg r <-> 4 consecutive n, lane&15 <-> m  (row-contiguous C stores)
#pragma unroll
            for (int ks = 0; ks < ST::KSTEPS; ++ks)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = MM::mma(fb[ks][j], fa[ks][i], acc[i][j]);
        }
    }
    stamp(g, 3);
    if constexpr (DIRECT_OK) {
        if (direct) {                   // registers -> global: nothing to wait for but this wave's own MFMAs
            u32x4 epre[2 * MI];
#pragma unroll
            for (int i = 0; i < 2 * MI; ++i) epre[i] = __builtin_bit_cast(u32x4, zpre[i]);
            gemm_epilogue_direct<MI>(g, acc, w, m0, n0, lane, epre);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the ring's look-ahead pieces past kend: 16 zero bytes per lane)
#ifdef A8_STAMPS
            __syncthreads();
            stamp(g, 4);
#endif
            return;
        }
    }
    // no DMA may still be in flight, and every wave must be done reading the ring, before LDS is reused
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    gemm_epilogue<T, MI, NWM, NP>(g, acc, lds, w, m0, n0, lane, zpre, use_zpre, bpre, use_bpre);
#ifdef A8_STAMPS
    __syncthreads();
    stamp(g, 4);
#endif
}

template <typename T, bool AKS, bool BKS, int KB, int NST, int NWM, bool A3 = false, bool DIRECT = false, bool KFAST = false>
A8_DEV void gemm_body(const GemmArgs& g, const int bid) {
    const int split = bid / (g.tiles_m * g.tiles_n);
    const int kbeg = split * g.kchunk;
    gemm_core<T, AKS, BKS, KB, NST, NWM, A3, DIRECT, KFAST>(g, bid - split * (g.tiles_m * g.tiles_n), kbeg, min(g.K, kbeg + g.kchunk));
}

// ---- the default 128 x 128 NT tile with the asymmetric ring (A three deep, B two deep: gemm_body, A3)
template <typename T, bool KFAST = false>
__global__ __launch_bounds__(512, 4) void gemm_a3_kernel(const GemmArgs g) {
    gemm_body<T, false, false, 2, 2, 4, true, true, KFAST>(g, xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n * g.nsplit));
}

// the asymmetric ring for the weight-gradient layout (both operands K-strided; a8_set_tuning key 15 = 0 restores the symmetric ring):
// bit-identical, the encoder layer's grouped weight gradients 123 -> 111 us
template <typename T>
__global__ __launch_bounds__(512, 4) void gemm_a3_tn_kernel(const GemmArgs g) {
    gemm_body<T, true, true, 2, 2, 4, true>(g, xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n * g.nsplit));
}

template <typename T, bool AKS, bool BKS, int KB, int NST, int NWM>
__global__ __launch_bounds__(128 * NWM, (NST * 2 * 8192 * KB <= 65536) ? NWM : NWM / 2) void gemm_kernel(const GemmArgs g) {
    gemm_body<T, AKS, BKS, KB, NST, NWM>(g, xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n * g.nsplit));
}

// several problems, one grid: block ids [start[i], start[i+1]) belong to problem i
constexpr int MAXGROUP = 8;
struct GemmGroup { int n; int start[MAXGROUP + 1]; GemmArgs g[MAXGROUP]; };
template <typename T, bool AKS, bool BKS, int KB, int NST, int NWM>
__global__ __launch_bounds__(128 * NWM, (NST * 2 * 8192 * KB <= 65536) ? NWM : NWM / 2) void gemm_grouped_kernel(const GemmGroup grp) {
    const int bid = xcd_remap(blockIdx.x, grp.start[grp.n]);
    int i = 0;
    while (i + 1 < grp.n && bid >= grp.start[i + 1]) ++i;
    const GemmArgs g = grp.g[i];                       // block-uniform: lives in SGPRs
    gemm_body<T, AKS, BKS, KB, NST, NWM>(g, bid - grp.start[i]);
}

template <typename T>
__global__ __launch_bounds__(512, 4) void gemm_grouped_a3_tn_kernel(const GemmGroup grp) {
    const int bid = xcd_remap(blockIdx.x, grp.start[grp.n]);
    int i = 0;
    while (i + 1 < grp.n && bid >= grp.start[i + 1]) ++i;
    const GemmArgs g = grp.g[i];
    gemm_body<T, true, true, 2, 2, 4, true>(g, bid - grp.start[i]);
}

// the grouped launch with the asymmetric ring of gemm_a3_kernel (bf16 NT; the conv data-gradient halves)
template <typename T, bool KFAST = false>
__global__ __launch_bounds__(512, 4) void gemm_grouped_a3_kernel(const GemmGroup grp) {
    const int bid = xcd_remap(blockIdx.x, grp.start[grp.n]);
    int i = 0;
    while (i + 1 < grp.n && bid >= grp.start[i + 1]) ++i;
    const GemmArgs g = grp.g[i];
    gemm_body<T, false, false, 2, 2, 4, true, true, KFAST>(g, bid - grp.start[i]);
}

// bf16: 128 B of K per row, 2-deep ring, 8 waves (4 waves/SIMD at 2 blocks/CU);  f32: 64-byte rows, 4-deep ring, 4 waves
template <typename T> struct Cfg;
template <> struct Cfg<bf16> { static constexpr int KB = 2, NST = 2, NWM = 4; };
template <> struct Cfg<float> { static constexpr int KB = 1, NST = 4, NWM = 2; };

template <typename T> int launch(const a8_gemm_desc* d, const GemmArgs& g, int grid, hipStream_t s) {
    typedef Cfg<T> C;
    const dim3 blk(128 * C::NWM);
    switch (d->layout) {
        case A8_GEMM_NT: hipLaunchKernelGGL((gemm_kernel<T, false, false, C::KB, C::NST, C::NWM>), dim3(grid), blk, 0, s, g); break;
        case A8_GEMM_NN: hipLaunchKernelGGL((gemm_kernel<T, false, true, C::KB, C::NST, C::NWM>), dim3(grid), blk, 0, s, g); break;
        case A8_GEMM_TN: hipLaunchKernelGGL((gemm_kernel<T, true, true, C::KB, C::NST, C::NWM>), dim3(grid), blk, 0, s, g); break;
        default: return a8_set_error(A8_ERR_ARG, "a8_gemm: bad layout %d", d->layout);
    }
    return a8_check_launch("a8_gemm");
}
template <typename T> int launch_grouped(int layout, const GemmGroup& g, int grid, hipStream_t s) {
    typedef Cfg<T> C;
    const dim3 blk(128 * C::NWM);
    switch (layout) {
        case A8_GEMM_NT: hipLaunchKernelGGL((gemm_grouped_kernel<T, false, false, C::KB, C::NST, C::NWM>), dim3(grid), blk, 0, s, g); break;
        case A8_GEMM_NN: hipLaunchKernelGGL((gemm_grouped_kernel<T, false, true, C::KB, C::NST, C::NWM>), dim3(grid), blk, 0, s, g); break;
        case A8_GEMM_TN: hipLaunchKernelGGL((gemm_grouped_kernel<T, true, true, C::KB, C::NST, C::NWM>), dim3(grid), blk, 0, s, g); break;
        default: return a8_set_error(A8_ERR_ARG, "a8_gemm_grouped: bad layout %d", layout);
    }
    return a8_check_launch("a8_gemm_grouped");
}

}  // namespace

#ifdef A8_STAMPS
static unsigned long long* g_stamps = nullptr;
extern "C" int a8_debug_set_stamps(void* p) { g_stamps = (unsigned long long*)p; return A8_OK; }
#endif

// validates one descriptor and fills the kernel arguments; *grid = blocks it needs
static int build_args(const a8_gemm_desc* d, GemmArgs& g, long long* grid) {
    A8_REQUIRE(d != nullptr, "a8_gemm: null descriptor");
    A8_REQUIRE(d->size == (int)sizeof(a8_gemm_desc), "a8_gemm: descriptor size %d, this library's a8_gemm_desc has %d bytes (binding built against another a8hip.h)",
               d->size, (int)sizeof(a8_gemm_desc));
    A8_REQUIRE(d->dtype == A8_F32 || d->dtype == A8_BF16, "a8_gemm: bad dtype %d", d->dtype);
    A8_REQUIRE(d->M >= 0 && d->N > 0 && d->K >= 0, "a8_gemm: bad sizes M=%d N=%d K=%d", d->M, d->N, d->K);
    const int esz = d->dtype == A8_BF16 ? 2 : 4, epc = 16 / esz;
    const bool aks = d->layout == A8_GEMM_TN, bks = d->layout != A8_GEMM_NT;
    A8_REQUIRE(d->N % 8 == 0 || (d->N % 4 == 0 && (d->out_f32 || d->dtype == A8_F32)), "a8_gemm: N=%d must be a multiple of 8 (4 for f32 output)", d->N);
    A8_REQUIRE(d->out_f32 || d->dtype == A8_F32 || (d->ldc % 8 == 0 && (!d->zout && !d->zin || d->ldz % 8 == 0)), "a8_gemm: ldc/ldz must be multiples of 8 for 16-bit output");
    if (!aks || !bks) A8_REQUIRE(d->K % epc == 0, "a8_gemm: K=%d must be a multiple of %d for a K-contiguous operand", d->K, epc);
    if (aks) A8_REQUIRE(d->M % epc == 0, "a8_gemm: M=%d must be a multiple of %d (TN)", d->M, epc);
    if (bks) A8_REQUIRE(d->N % epc == 0, "a8_gemm: N=%d must be a multiple of %d (NN/TN)", d->N, epc);
    A8_REQUIRE(d->lda % epc == 0 && d->ldb % epc == 0 && d->ldc % 4 == 0, "a8_gemm: leading dimensions must keep 16-byte alignment");
    A8_REQUIRE(((uintptr_t)d->A % 16) == 0 && ((uintptr_t)d->B % 16) == 0 && ((uintptr_t)d->C % 16) == 0, "a8_gemm: pointers must be 16-byte aligned");
    A8_REQUIRE(d->a_bs % epc == 0 && d->b_bs % epc == 0, "a8_gemm: batch strides must keep 16-byte alignment");
    const int bk = d->dtype == A8_BF16 ? 64 : 16;
    A8_REQUIRE(d->a_rpb >= 0 && d->b_rpb >= 0 && d->c_rpb >= 0 && d->z_rpb >= 0 && d->a_rpb < (1ll << 31) && d->b_rpb < (1ll << 31) &&
               d->c_rpb < (1ll << 31) && d->z_rpb < (1ll << 31), "a8_gemm: rows-per-batch remaps must fit 31 bits");
    g.A = Op{(const char*)d->A, d->lda, d->a_rpb, d->a_bs};
    g.B = Op{(const char*)d->B, d->ldb, d->b_rpb, d->b_bs};
    g.C = (char*)d->C; g.ldc = d->ldc; g.c_rpb = d->c_rpb; g.c_bs = d->c_bs;
    g.zin = (const char*)d->zin; g.zout = (char*)d->zout; g.ldz = d->ldz; g.z_rpb = d->z_rpb; g.z_bs = d->z_bs;
    g.bias = d->bias; g.bias_rows = d->bias_rows; g.colsum = d->colsum;
    g.M = d->M; g.N = d->N; g.K = d->K;
    g.tiles_m = cdiv(d->M, 128); g.tiles_n = cdiv(d->N, 128);
    int nsplit = d->splitk > 0 ? d->splitk : 1;
    int kchunk = cdiv(cdiv(d->K, nsplit), bk) * bk;
    if (kchunk == 0) kchunk = bk;
    nsplit = d->K > 0 ? cdiv(d->K, kchunk) : 1;
    g.kchunk = kchunk; g.nsplit = nsplit;
    g.act = d->act; g.out_f32 = d->out_f32; g.alpha = d->alpha;
    g.group_m = 8;                                                   // m-tiles per tile-order group (1..16 measured: 8)
    g.direct_off = g_a8_tuning[22] == 1;                             // (A/B: LDS-staged epilogue in the NT kernels too)
#ifdef A8_STAMPS
    g.stamps = g_stamps;
#endif
    g.atomic = (nsplit > 1 || d->accumulate) ? 1 : 0;
    A8_REQUIRE(!(g.atomic && !d->out_f32), "a8_gemm: split-K / accumulate needs an f32 output");
    A8_REQUIRE(!(nsplit > 1 && (d->bias || d->act || d->zout || d->colsum)), "a8_gemm: split-K cannot carry a bias/activation/colsum epilogue");
    A8_REQUIRE((d->act != A8_ACT_MUL_GELU_GRAD && d->act != A8_ACT_ADD) || d->zin, "a8_gemm: MUL_GELU_GRAD / ADD need zin");
    *grid = (long long)g.tiles_m * g.tiles_n * nsplit;
    A8_REQUIRE(*grid < (1ll << 30), "a8_gemm: grid too large");
    return A8_OK;
}

extern "C" int a8_gemm(const a8_gemm_desc* d, void* stream) {
    GemmArgs g;
    long long grid = 0;
    if (int e = build_args(d, g, &grid)) return e;
    if (d->M == 0) return A8_OK;
    hipStream_t s = (hipStream_t)stream;
    // many rows, N % 256 == 0: BM x 256 tiles on persistent blocks (gemm_stream.hip; a8_set_tuning key 5 = minimum M, 0 = never)
    if (const int r = a8i_gemm_stream_try(d, stream)) return r > 0 ? A8_OK : -r;
    // Asymmetric ring (a8_set_tuning key 14: 0 = never, otherwise the minimum M; 1 = every bf16 NT problem)
    if (d->dtype == A8_BF16 && d->layout == A8_GEMM_NT && g_a8_tuning[14] > 0 && d->M >= g_a8_tuning[14]) {
        // at least one whole k-tile, whole k-tiles in every split and M, N >= 1: the lean-addressing instantiation (a8_set_tuning key 25 = 1: never).
        // K = 0 (an empty reduction: C = bias / epilogue of zero) has no k-tile to clamp to and takes the predicate form, which reads the zero line
        if (g.K >= 64 && g.K % 64 == 0 && g.kchunk % 64 == 0 && d->M >= 1 && g_a8_tuning[25] == 0) hipLaunchKernelGGL((gemm_a3_kernel<bf16, true>), dim3((int)grid), dim3(512), 0, s, g);
        else hipLaunchKernelGGL((gemm_a3_kernel<bf16, false>), dim3((int)grid), dim3(512), 0, s, g);
        return a8_check_launch("a8_gemm");
    }
    if (d->dtype == A8_BF16 && d->layout == A8_GEMM_TN && g_a8_tuning[15] > 0) {
        hipLaunchKernelGGL(gemm_a3_tn_kernel<bf16>, dim3((int)grid), dim3(512), 0, s, g);
        return a8_check_launch("a8_gemm");
    }
    return d->dtype == A8_BF16 ? launch<bf16>(d, g, (int)grid, s) : launch<float>(d, g, (int)grid, s);
}

extern "C" int a8_gemm_grouped(const a8_gemm_desc* descs, int n, void* stream) {
    A8_REQUIRE(descs != nullptr && n >= 1 && n <= MAXGROUP, "a8_gemm_grouped: need 1..%d descriptors (got %d)", MAXGROUP, n);
    GemmGroup grp;
    grp.n = 0;
    grp.start[0] = 0;
    long long total = 0;
    for (int i = 0; i < n; ++i) {
        A8_REQUIRE(descs[i].dtype == descs[0].dtype && descs[i].layout == descs[0].layout, "a8_gemm_grouped: all problems must share dtype and layout");
        long long grid = 0;
        if (int e = build_args(&descs[i], grp.g[grp.n], &grid)) return e;
        if (descs[i].M == 0) continue;
        total += grid;
        A8_REQUIRE(total < (1ll << 30), "a8_gemm_grouped: grid too large");
        grp.start[++grp.n] = (int)total;
    }
    if (grp.n == 0) return A8_OK;
    // every problem fit for the stream kernel (the conv data gradients' even / odd halves: each fills the chip for several rounds
    // by itself): one persistent launch each instead of the shared grid
    if (descs[0].dtype == A8_BF16 && descs[0].layout == A8_GEMM_NT && a8i_gemm_stream_eligible(&descs[0])) {
        bool all = true;
        for (int i = 1; i < n && all; ++i) all = descs[i].M == 0 || a8i_gemm_stream_eligible(&descs[i]);
        if (all) {
            for (int i = 0; i < n; ++i) {
                if (descs[i].M == 0) continue;
                const int r = a8i_gemm_stream_try(&descs[i], stream);
                if (r < 0) return -r;
                A8_REQUIRE(r == 1, "a8_gemm_grouped: stream kernel refused an eligible problem");
            }
            return A8_OK;
        }
    }
    hipStream_t s = (hipStream_t)stream;
    if (descs[0].dtype == A8_BF16) {
        bool a3 = descs[0].layout == A8_GEMM_NT && g_a8_tuning[14] > 0;
        for (int i = 0; i < n && a3; ++i) a3 = descs[i].M == 0 || descs[i].M >= g_a8_tuning[14];
        if (a3) {
            bool kfast = g_a8_tuning[25] == 0;
            for (int i = 0; i < grp.n && kfast; ++i) kfast = grp.g[i].K >= 64 && grp.g[i].K % 64 == 0 && grp.g[i].kchunk % 64 == 0;      // (K = 0: nkb = 0, the lean form would address k-tile -1)
            if (kfast) hipLaunchKernelGGL((gemm_grouped_a3_kernel<bf16, true>), dim3((int)total), dim3(512), 0, s, grp);
            else hipLaunchKernelGGL((gemm_grouped_a3_kernel<bf16, false>), dim3((int)total), dim3(512), 0, s, grp);
            return a8_check_launch("a8_gemm_grouped");
        }
        if (descs[0].layout == A8_GEMM_TN && g_a8_tuning[15] > 0) {
            hipLaunchKernelGGL(gemm_grouped_a3_tn_kernel<bf16>, dim3((int)total), dim3(512), 0, s, grp);
            return a8_check_launch("a8_gemm_grouped");
        }
        return launch_grouped<bf16>(descs[0].layout, grp, (int)total, s);
    }
    return launch_grouped<float>(descs[0].layout, grp, (int)total, s);
}
