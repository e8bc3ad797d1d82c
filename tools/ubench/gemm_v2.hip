// Standalone experiment (not part of liba8hip): the round-4 "stream" GEMM for bf16 NT products with N % 256 == 0, K % 128 == 0.
//
//   * tile = BM x 256 with BM = 128 + 16 f (f = 1..8) chosen per launch so that the tiles fill whole rounds of the 256 CUs
//     (8 waves: wave row 0 always owns 128 rows, wave row 1 owns the remaining 16 f; one wave of each row per SIMD, so the SIMDs
//     stay balanced);
//   * persistent blocks, ONE continuous k-tile stream across a block's tiles: A three k-tiles deep in a 3-slot ring (activations:
//     cold, from HBM), B two deep in a 2-slot ring (weights: L2 resident) = 160 KB of LDS; the next tile's first k-tiles are in
//     flight while the current tile finishes and stores;
//   * the pipelined quadrant schedule of gemm_pipe.hip (fragment reads of the next quadrant issued ahead of the MFMAs of the
//     current one, one barrier per k-tile placed mid-tile);
//   * every wait is "all but my 4 youngest DMA pieces" (never a count that depends on the epilogue's stores): one extra wait in
//     front of the epilogue retires the next tile's first operands, the first k-tile of a tile then only needs the barrier;
//   * epilogue without LDS: two v_permlane16_swap per fragment pair turn the 8-byte column groups of the accumulator layout into
//     16-byte stores (16 rows x 64 contiguous bytes per wave-instruction).
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -I audio8_amd/csrc tools/ubench/gemm_v2.hip -o /tmp/gemm_v2 && /tmp/gemm_v2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <type_traits>

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define DEV __device__ __forceinline__
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int BN = 256, BK = 64, KROW = 128;
constexpr int TA = 256 * KROW, TB = BN * KROW;                  // 32 KB each
constexpr int RING = 3 * TA + 2 * TB;                           // 160 KB

__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

DEV void glds16(const char* src, char* dst_wave_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst_wave_uniform, 16, 0, 0);
}
DEV bf16x8 frag(const char* tile, int rb, int ks, int lane) {     // 16 rows x 32 k: lane&15 = row, lane>>4 = 16-byte chunk of the k-step
    const int row = rb + (lane & 15);
    return *reinterpret_cast<const bf16x8*>(tile + row * KROW + (((ks * 4 + (lane >> 4)) ^ (row & 7)) << 4));
}
DEV int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}
// exact-erf GELU of the product (a8_common.h), two elements at a time
DEV f32x2 splat2(float c) { return (f32x2){c, c}; }
DEV f32x2 norm_cdf2(f32x2 x) {
    const f32x2 t = {fminf(fabsf(x[0]), 5.75f), fminf(fabsf(x[1]), 5.75f)};
    f32x2 r = splat2(8.510881681e-06f);
    r = __builtin_elementwise_fma(r, t, splat2(-5.449422293e-05f));
    r = __builtin_elementwise_fma(r, t, splat2(-4.180608761e-04f));
    r = __builtin_elementwise_fma(r, t, splat2(7.381212995e-03f));
    r = __builtin_elementwise_fma(r, t, splat2(-5.268092407e-02f));
    r = __builtin_elementwise_fma(r, t, splat2(-4.591593132e-01f));
    r = __builtin_elementwise_fma(r, t, splat2(-1.151109407e+00f));
    const f32x2 rt = r * t;
    const f32x2 e = {__builtin_amdgcn_exp2f(rt[0]), __builtin_amdgcn_exp2f(rt[1])};
    const f32x2 h = __builtin_elementwise_fma(splat2(-0.5f), e, splat2(0.5f));
    return (f32x2){0.5f + copysignf(h[0], x[0]), 0.5f + copysignf(h[1], x[1])};
}
DEV unsigned pack2(float a, float b) {
    bf16x2 v = {(bf16)a, (bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

struct V2Args {
    const bf16* A; int lda, a_rpb, a_bs;             // row m of A starts at (m / a_rpb) * a_bs + (m % a_rpb) * lda elements (a_rpb = 0: m * lda)
    const bf16* B; int ldb;
    bf16* C; bf16* Z; int ldc;
    const float* bias;
    int M, N, K, BM, tiles_m, tiles_n;
    int rpb, tpb;                                     // rows per batch item (M = batches * rpb), tiles per batch item
    int stagger;                                      // experiment: block b starts (b & 3) * stagger ticks of 10 ns late
};

DEV long long row_off(const V2Args& g, int m) {
    if (!g.a_rpb) return (long long)m * g.lda;
    const unsigned q = (unsigned)m / (unsigned)g.a_rpb, r = (unsigned)m - q * (unsigned)g.a_rpb;
    return (long long)q * g.a_bs + (long long)r * g.lda;
}

// EPI 0: C = bf16(acc + bias).  EPI 1: Z = bf16(acc + bias), C = bf16(gelu(acc + bias)).
// MI = 16-row fragments per wave (5..8): tile height BM = 32 MI, wave row wr owns rows [16 MI wr, 16 MI (wr + 1)).
// Needs K % 128 == 0, K >= 256, N % 256 == 0, rows per batch item >= 1.
struct TileAt { unsigned a, b; int row, n0; };       // byte offsets of the tile's first A row / B row, first output row, first column      // wave-uniform description of one tile

template <int MI, int EPI, int NA>
__global__ __launch_bounds__(512, 2) void gemm_v2(const V2Args g) {
    static_assert(NA == 2 || NA == 3, "A ring depth");
    constexpr int BM = 32 * MI, WR = 16 * MI;
    constexpr int MI1 = MI - 4;                               // fragments in the second A quarter-read (the first always has 4)
    __shared__ __attribute__((aligned(1024))) char lds[NA * TA + 2 * TB];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6), wr = w >> 2, wc = w & 3;
    const int total = g.tiles_m * g.tiles_n, nk = g.K / BK, G = gridDim.x;
    constexpr int GROUP_M = 8;
    const char* const Ab = reinterpret_cast<const char*>(g.A);
    const char* const Bb = reinterpret_cast<const char*>(g.B);
    // tile v -> batch item, first row inside it (a tile never straddles two batch items), first column; scalar arithmetic only.
    // Past the end of the block's sequence the last tile is described again: the stream keeps issuing (valid, unused) pieces,
    // so every wave issues the same number of pieces at every position and one counted wait serves the whole kernel.
    auto tile_at = [&](int v) {
        TileAt d;
        const int tile = xcd_remap(min(v, total - 1 - ((total - 1 - (int)blockIdx.x) % G)), total);
        const int tpg = GROUP_M * g.tiles_n, grp = tile / tpg, first_m = grp * GROUP_M;
        const int gsz = min(GROUP_M, g.tiles_m - first_m), tin = tile - grp * tpg;
        const int mt = first_m + tin % gsz;
        const int bi = mt / g.tpb;
        const int r0 = min((mt - bi * g.tpb) * BM, g.rpb - BM);      // the last tile of a batch item ends with it (rows of its neighbour recomputed)
        d.n0 = (tin / gsz) * BN;
        d.row = bi * g.rpb + r0;
        d.a = ((unsigned)bi * (unsigned)g.a_bs + (unsigned)r0 * (unsigned)g.lda) * 2u;       // (spans < 4 GB: checked on the host)
        d.b = (unsigned)d.n0 * (unsigned)g.ldb * 2u;
        return d;
    };

    // ---- the DMA cursors.  32 one-KB pieces (8 rows) per operand per k-tile, 4 per wave (rows 64 apart); XOR swizzle on the source.
    // A piece's source = [tile base + 64 j rows + k offset] (scalar) + [lane's row * row stride + swizzled chunk] (one register):
    // every tile has BM valid rows (the last tile of a batch item is shifted up to end with it), pieces below the tile (wave-uniform)
    // re-read piece 0.  LDS map: B slots at 0 / 32 KB, A slots at 64 / 96 / 128 KB.
    const int srow = 8 * w + (lane >> 3);
    const unsigned sch16 = (unsigned)(((lane & 7) ^ (srow & 7)) << 4);
    const unsigned arow = (unsigned)(srow * g.lda * 2) + sch16;
    const unsigned brow = (unsigned)(srow * g.ldb * 2) + sch16;
    const unsigned ajstep = 64u * g.lda * 2u, bjstep = 64u * g.ldb * 2u;
    TileAt cur = tile_at(blockIdx.x), nxt = tile_at(blockIdx.x + G);
    const char* a_base = Ab + cur.a; int a_kt = 0;
    const char* b_base = Bb + cur.b; int b_kt = 0;
    auto issue_a = [&](int slot) {
        char* ta = lds + 2 * TB + slot * TA;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const char* sj = a_base + ((8 * w + 64 * j < BM) ? j * ajstep : 0);       // wave-uniform
            glds16(sj + arow, ta + (w + 8 * j) * 1024);
        }
        a_base += KROW;
        if (++a_kt == nk) { a_kt = 0; a_base = Ab + nxt.a; }
    };
    auto issue_b = [&](int slot) {
        char* tb = lds + slot * TB;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16((b_base + j * bjstep) + brow, tb + (w + 8 * j) * 1024);
        b_base += KROW;
        if (++b_kt == nk) { b_kt = 0; b_base = Bb + nxt.b; }
    };

    f32x4 acc[MI][4];
    bf16x8 a0[2][4], a1[2][MI1], b0[2][2], b1[2][2];
    const int ra = wr * WR, cb = wc * 64;
    // Fragment reads.  A and B tiles share one image ([row][128 B], 16-byte chunk c of row r stored at c ^ (r & 7)), so a lane's
    // byte offset inside ANY 16-row fragment is one constant; the two k-steps differ by XOR 64.  Two lane constants in all (with the
    // wave's first row / column folded in); a slot's base is formed next to its reads by an opaque add (so that the compiler cannot
    // hoist five slots x two k-steps x two operands of loop-invariant addresses out of the loop and spill them), everything else is
    // the instruction's immediate offset.
    typedef const __attribute__((address_space(3))) bf16x8* LdsFrag;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned pat = lds0 + (unsigned)((lane & 15) * KROW + (((lane >> 4) ^ (lane & 7)) << 4));
    const unsigned pat_a = pat + (unsigned)(ra * KROW), pat_b = pat + (unsigned)(cb * KROW);
    auto slot_base = [&](unsigned p, int off) {
        unsigned r;
        asm volatile("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(p), "s"(off));
        return r;
    };
    auto rd_a0 = [&](int slot_off) {
        const unsigned p0 = slot_base(pat_a, slot_off), p1 = p0 ^ 64u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a0[0][i] = *(LdsFrag)(size_t)(p0 + 16 * i * KROW);
            a0[1][i] = *(LdsFrag)(size_t)(p1 + 16 * i * KROW);
        }
    };
    auto rd_a1 = [&](int slot_off) {
        const unsigned p0 = slot_base(pat_a, slot_off), p1 = p0 ^ 64u;
#pragma unroll
        for (int i = 0; i < MI1; ++i) {
            a1[0][i] = *(LdsFrag)(size_t)(p0 + (64 + 16 * i) * KROW);
            a1[1][i] = *(LdsFrag)(size_t)(p1 + (64 + 16 * i) * KROW);
        }
    };
    auto rd_b = [&](bf16x8 (&f)[2][2], int slot_off, int half) {
        const unsigned p0 = slot_base(pat_b, slot_off), p1 = p0 ^ 64u;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f[0][i] = *(LdsFrag)(size_t)(p0 + (32 * half + 16 * i) * KROW);
            f[1][i] = *(LdsFrag)(size_t)(p1 + (32 * half + 16 * i) * KROW);
        }
    };
    // swapped operands: accumulator register r <-> 4 consecutive n, lane & 15 <-> m
    auto mma0 = [&](const bf16x8 (&fb)[2][2], int bh) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][2 * bh + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][j], a0[ks][i], acc[i][2 * bh + j], 0, 0, 0);
    };
    auto mma1 = [&](const bf16x8 (&fb)[2][2], int bh) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < MI1; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[4 + i][2 * bh + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][j], a1[ks][i], acc[4 + i][2 * bh + j], 0, 0, 0);
    };

    // one stream position with A in ring slot SA, B in slot SB (compile-time: every LDS address is a lane constant + an immediate).
    // On entry a0 / bx hold its first A / B quarters, by is free; on exit a0 / by hold the first quarters of the next position
    // (read while the last quadrant's MFMAs run): the two B register sets swap roles every position.
    auto step = [&](auto SA_, auto SB_, const bool wait, bf16x8 (&bx)[2][2], bf16x8 (&by)[2][2]) __attribute__((always_inline)) {
        constexpr int SA = decltype(SA_)::value, SB = decltype(SB_)::value, SA1 = (SA + 1) % NA;
        constexpr int ta = 2 * TB + SA * TA, tb = SB * TB, na = 2 * TB + SA1 * TA, nb = (SB ^ 1) * TB;
        rd_b(by, tb, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma0(bx, 0);
        __builtin_amdgcn_sched_barrier(0);
        rd_a1(ta);
        __builtin_amdgcn_sched_barrier(0);
        mma0(by, 1);
        __builtin_amdgcn_sched_barrier(0);
        // all but my 4 youngest pieces have landed (the A k-tile issued last stays in flight)
        if (wait) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA == 3 ? 4 : 0) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue_b(SB);                                     // position + 2 -> the B slot just freed
        issue_a(SA);                                     // position + 3 -> the A slot just freed
        __builtin_amdgcn_sched_barrier(0);
        mma1(by, 1);
        __builtin_amdgcn_sched_barrier(0);
        rd_a0(na);
        rd_b(by, nb, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma1(bx, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    typedef std::integral_constant<int, 0> I0; typedef std::integral_constant<int, 1> I1; typedef std::integral_constant<int, 2> I2;
    if (g.stagger) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), dt = (unsigned long long)((blockIdx.x >> 3) & 3) * g.stagger;
        while (__builtin_amdgcn_s_memrealtime() - t0 < dt) __builtin_amdgcn_s_sleep(8);
    }
    // ---- prologue: positions 0, 1 (A and B) and 2 (A)
    issue_b(0); issue_a(0); issue_b(1); issue_a(1);
    if (NA == 3) issue_a(2);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA == 3 ? 4 : 0) : "memory");
    __builtin_amdgcn_s_barrier();
    rd_a0(2 * TB);
    rd_b(b0, 0, 0);
    for (int v = blockIdx.x; v < total; v += G) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // One turn of both rings = 6 positions (NA = 3: K % 384 == 0, so every tile starts at phase 0) or 2 (NA = 2: K % 128 == 0).
        // A tile's first position skips the DMA wait: the one in front of the previous epilogue (or the prologue's) covers it.
        if constexpr (NA == 3) {
            for (int kt = 0; kt < nk; kt += 6) {
                step(I0{}, I0{}, kt != 0, b0, b1);
                step(I1{}, I1{}, true, b1, b0);
                step(I2{}, I0{}, true, b0, b1);
                step(I0{}, I1{}, true, b1, b0);
                step(I1{}, I0{}, true, b0, b1);
                step(I2{}, I1{}, true, b1, b0);
            }
        } else {
            for (int kt = 0; kt < nk; kt += 2) {
                step(I0{}, I0{}, kt != 0, b0, b1);
                step(I1{}, I1{}, true, b1, b0);
            }
        }
        // my pieces of the next tile's k-tile 1 have landed before any store is issued: its first step needs only the barrier
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA == 3 ? 4 : 0) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        // both cursors are inside the next tile by now (K >= 256): describe the tile after it
        const TileAt done = cur;
        cur = nxt;
        nxt = tile_at(v + 2 * G);
        // ---- epilogue: acc[i][j] register r = column 16 j + 4 (lane >> 4) + r of row 16 i + (lane & 15)
        const int li = lane & 15, gq = lane >> 4;
        bf16* const crow0 = g.C + (long long)done.row * g.ldc;
        bf16* const zrow0 = g.Z + (long long)done.row * g.ldc;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            float bz[2][4];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) bz[q][e] = g.bias ? g.bias[done.n0 + cb + 32 * jj + 16 * q + 4 * gq + e] : 0.f;
            // lanes of 16-lane row gq: even gq keeps fragment 2jj (its own 4 columns + the next row's), odd gq fragment 2jj+1
            const int col = done.n0 + cb + 32 * jj + 16 * (gq & 1) + 8 * (gq >> 1);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int rr = ra + 16 * i + li;
                const unsigned coff = (unsigned)rr * (unsigned)g.ldc + (unsigned)col;
                unsigned zl[2], zh[2], yl[2], yh[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    f32x4 x = acc[i][2 * jj + q];
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] += bz[q][e];
                    constexpr bool GELU = EPI == 1 || EPI == 2, ZOUT = EPI == 1 || EPI == 3 || EPI == 4;     // (2, 3: cost-splitting experiments)
                    if (ZOUT) { zl[q] = pack2(x[0], x[1]); zh[q] = pack2(x[2], x[3]); }
                    if (GELU) {
                        const f32x2 p0 = {x[0], x[1]}, p1 = {x[2], x[3]};
                        const f32x2 y0 = p0 * norm_cdf2(p0), y1 = p1 * norm_cdf2(p1);
                        yl[q] = pack2(y0[0], y0[1]); yh[q] = pack2(y1[0], y1[1]);
                    } else {
                        yl[q] = pack2(x[0], x[1]); yh[q] = pack2(x[2], x[3]);
                    }
                }
                {
                    auto s0 = __builtin_amdgcn_permlane16_swap(yl[0], yl[1], false, false);
                    auto s1 = __builtin_amdgcn_permlane16_swap(yh[0], yh[1], false, false);
                    const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                    *reinterpret_cast<u32x4*>(crow0 + coff) = o;
                }
                if (EPI == 1 || EPI == 3 || EPI == 4) {
                    auto s0 = __builtin_amdgcn_permlane16_swap(zl[0], zl[1], false, false);
                    auto s1 = __builtin_amdgcn_permlane16_swap(zh[0], zh[1], false, false);
                    const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                    *reinterpret_cast<u32x4*>((EPI == 4 ? crow0 : zrow0) + coff) = o;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- the round-2/3 pipelined kernel (gemm_p256_kernel of csrc/gemm.hip), for the same-process comparison: 2-deep symmetric ring,
// one block per tile, 8-byte stores straight from the accumulators
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_p256(const V2Args g) {
    constexpr int STAGE = TA + TB;
    __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6), wr = w >> 2, wc = w & 3;
    const int tile = xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
    constexpr int GROUP_M = 8;
    const int tpg = GROUP_M * g.tiles_n, grp = tile / tpg, first_m = grp * GROUP_M;
    const int gsz = min(GROUP_M, g.tiles_m - first_m), tin = tile - grp * tpg;
    const int m0 = (first_m + tin % gsz) * 256, n0 = (tin / gsz) * 256;
    const int nk = g.K / 64;
    const int srow = 8 * w + (lane >> 3), sch = (lane & 7) ^ (srow & 7);
    unsigned a_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = min(m0 + srow + 64 * j, g.M - 1);
        a_off[j] = (unsigned)(row_off(g, row) * 2 + sch * 16);
    }
    const unsigned b_off = (unsigned)((long long)(n0 + srow) * g.ldb * 2 + sch * 16);
    const unsigned bjstep = (unsigned)(64 * g.ldb * 2);
    unsigned koff = 0;
    const char* const Ab = reinterpret_cast<const char*>(g.A);
    const char* const Bb = reinterpret_cast<const char*>(g.B);
    auto issue = [&](int kt) {
        char* ta = lds + (kt & 1) * STAGE;
        char* tb = ta + TA;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            glds16(Ab + (a_off[j] + koff), ta + (w + 8 * j) * 1024);
            glds16(Bb + (b_off + j * bjstep + koff), tb + (w + 8 * j) * 1024);
        }
        koff += KROW;
    };
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 a0[2][4], a1[2][4], b0[2][2], b1[2][2];
    const int ra = wr * 128, cb = wc * 64;
    auto rd_a = [&](bf16x8 (&f)[2][4], const char* ta, int half) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) f[ks][i] = frag(ta, ra + 64 * half + 16 * i, ks, lane);
    };
    auto rd_b = [&](bf16x8 (&f)[2][2], const char* tb, int half) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) f[ks][i] = frag(tb, cb + 32 * half + 16 * i, ks, lane);
    };
    auto mma = [&](const bf16x8 (&fa)[2][4], const bf16x8 (&fb)[2][2], int ah, int bh) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[4 * ah + i][2 * bh + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][j], fa[ks][i], acc[4 * ah + i][2 * bh + j], 0, 0, 0);
    };
    auto step = [&](int kt, bf16x8 (&bx)[2][2], bf16x8 (&by)[2][2]) {
        const char* ta = lds + (kt & 1) * STAGE;
        const char* tb = ta + TA;
        const char* na = lds + ((kt + 1) & 1) * STAGE;
        const char* nb = na + TA;
        rd_b(by, tb, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(a0, bx, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        rd_a(a1, ta, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(a0, by, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) issue(kt + 2);
        __builtin_amdgcn_sched_barrier(0);
        mma(a1, by, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) {
            rd_a(a0, na, 0);
            rd_b(by, nb, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma(a1, bx, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    issue(0);
    if (nk > 1) issue(1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8) : "memory");
    if (nk <= 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    rd_a(a0, lds, 0);
    rd_b(b0, lds + TA, 0);
    for (int kt = 0; kt < nk; kt += 2) {
        step(kt, b0, b1);
        if (kt + 1 < nk) step(kt + 1, b1, b0);
    }
    const int li = lane & 15, c4 = 4 * (lane >> 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + ra + 16 * i + li;
        if (m >= g.M) continue;
        const long long crow = (long long)m * g.ldc;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + cb + 16 * j + c4;
            f32x4 x = acc[i][j];
            if (EPI == 1) {
                bf16x4 z = {(bf16)x[0], (bf16)x[1], (bf16)x[2], (bf16)x[3]};
                *reinterpret_cast<bf16x4*>(g.Z + crow + n) = z;
                const f32x2 p0 = {x[0], x[1]}, p1 = {x[2], x[3]};
                const f32x2 y0 = p0 * norm_cdf2(p0), y1 = p1 * norm_cdf2(p1);
                bf16x4 o = {(bf16)y0[0], (bf16)y0[1], (bf16)y1[0], (bf16)y1[1]};
                *reinterpret_cast<bf16x4*>(g.C + crow + n) = o;
            } else {
                bf16x4 o = {(bf16)x[0], (bf16)x[1], (bf16)x[2], (bf16)x[3]};
                *reinterpret_cast<bf16x4*>(g.C + crow + n) = o;
            }
        }
    }
}

__global__ void ref_rows(const V2Args g, const int* rows, int nrows, float* R, int gelu) {      // f32 reference of selected rows
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = rows[blockIdx.y];
    if (n >= g.N) return;
    const bf16* a = g.A + row_off(g, m);
    float s = 0.f;
    for (int k = 0; k < g.K; ++k) s += (float)a[k] * (float)g.B[(long long)n * g.ldb + k];
    if (g.bias) s += g.bias[n];
    R[(long long)blockIdx.y * g.N + n] = s;
}
__global__ void fill(bf16* p, long long n, uint32_t seed, int integer) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 0x9E3779B1u + seed;
        h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
        p[i] = integer ? (bf16)(float)((int)(h % 5) - 2) : (bf16)(((float)(h >> 8) * (1.0f / 8388608.0f)) - 1.0f);
    }
}

static int best_bm(int M, int nb, int tiles_n, int* rounds_out) {      // the tile height that minimises (rounds x height), 256 CUs, one block each
    int best = 256; double bc = 1e30;
    for (int bm = 160; bm <= 256; bm += 32) {
        const long long tiles = (long long)nb * ((M / nb + bm - 1) / bm) * tiles_n;
        const long long rounds = (tiles + 255) / 256;
        const double cost = rounds * (bm + 160.0);            // MFMA time ~ bm, operand feed ~ bm + 256: weigh both
        if (cost < bc) { bc = cost; best = bm; if (rounds_out) *rounds_out = (int)rounds; }
    }
    return best;
}

struct Problem { const char* name; int M, N, K; int lda, rpb, bs; };

template <int EPI> static void launch_v2(V2Args g, int bm, int grid) {
    g.BM = bm; g.rpb = g.a_rpb ? g.a_rpb : g.M; g.tpb = (g.rpb + bm - 1) / bm;
    g.tiles_m = (g.M / g.rpb) * g.tpb; g.tiles_n = g.N / BN;
    if (!g.a_rpb) g.a_bs = 0;
    const int total = g.tiles_m * g.tiles_n;
    if (grid <= 0 || grid > total) grid = std::min(total, 256);
    const bool deep = (g.K / BK) % 6 == 0;
#define V2_LAUNCH(MI_) do { if (deep) hipLaunchKernelGGL((gemm_v2<MI_, EPI, 3>), dim3(grid), dim3(512), 0, 0, g); \
                            else hipLaunchKernelGGL((gemm_v2<MI_, EPI, 2>), dim3(grid), dim3(512), 0, 0, g); } while (0)
    switch (bm) {
        case 160: V2_LAUNCH(5); break;
        case 192: V2_LAUNCH(6); break;
        case 224: V2_LAUNCH(7); break;
        case 256: V2_LAUNCH(8); break;
        default: fprintf(stderr, "bad BM %d\n", bm); exit(1);
    }
#undef V2_LAUNCH
}
template <int EPI> static void launch_p256(V2Args g) {
    g.BM = 256; g.tiles_m = (g.M + 255) / 256; g.tiles_n = g.N / BN;
    hipLaunchKernelGGL(gemm_p256<EPI>, dim3(g.tiles_m * g.tiles_n), dim3(512), 0, 0, g);
}

int main(int argc, char** argv) {
    const bool quick = argc > 1 && atoi(argv[1]) == 1;
    // ---- correctness: integer operands (exact in f32), every row checked on small problems, sampled rows on big ones
    {
        struct T { int M, N, K, bm, grid, epi; };
        const T tests[] = {{1000, 512, 384, 256, 0, 0}, {1000, 512, 384, 160, 3, 0}, {3124, 768, 768, 192, 0, 1}, {3124, 768, 384, 160, 7, 1},
                           {777, 256, 384, 256, 1, 0}, {777, 256, 384, 224, 2, 0}, {2000, 512, 512, 224, 3, 1}, {2000, 512, 1024, 160, 2, 0}, {900, 256, 256, 192, 1, 1}, {5000, 512, 640, 256, 4, 1}, {6248, 2304, 768, 224, 0, 1}};
        for (const T& tc : tests) {
            bf16 *A, *B, *C, *Z; float *R, *bias; int* rows;
            CK(hipMalloc(&A, sizeof(bf16) * (size_t)tc.M * tc.K)); CK(hipMalloc(&B, sizeof(bf16) * (size_t)tc.N * tc.K));
            CK(hipMalloc(&C, sizeof(bf16) * (size_t)tc.M * tc.N)); CK(hipMalloc(&Z, sizeof(bf16) * (size_t)tc.M * tc.N));
            CK(hipMalloc(&R, sizeof(float) * (size_t)tc.M * tc.N)); CK(hipMalloc(&bias, sizeof(float) * tc.N)); CK(hipMalloc(&rows, sizeof(int) * tc.M));
            hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, 0, A, (long long)tc.M * tc.K, 1u, 1);
            hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, 0, B, (long long)tc.N * tc.K, 2u, 1);
            std::vector<float> hb(tc.N); for (int i = 0; i < tc.N; ++i) hb[i] = (float)((i * 7) % 5 - 2);
            std::vector<int> hrow(tc.M); for (int i = 0; i < tc.M; ++i) hrow[i] = i;
            CK(hipMemcpy(bias, hb.data(), sizeof(float) * tc.N, hipMemcpyHostToDevice)); CK(hipMemcpy(rows, hrow.data(), sizeof(int) * tc.M, hipMemcpyHostToDevice));
            V2Args g{A, tc.K, 0, 0, B, tc.K, C, Z, tc.N, bias, tc.M, tc.N, tc.K, 0, 0, 0, 0, 0, 0};
            hipLaunchKernelGGL(ref_rows, dim3((tc.N + 255) / 256, tc.M), dim3(256), 0, 0, g, rows, tc.M, R, 0);
            CK(hipMemset(C, 0xFF, sizeof(bf16) * (size_t)tc.M * tc.N)); CK(hipMemset(Z, 0xFF, sizeof(bf16) * (size_t)tc.M * tc.N));
            if (tc.epi) launch_v2<1>(g, tc.bm, tc.grid); else launch_v2<0>(g, tc.bm, tc.grid);
            CK(hipDeviceSynchronize());
            std::vector<bf16> hc((size_t)tc.M * tc.N), hz((size_t)tc.M * tc.N); std::vector<float> hr((size_t)tc.M * tc.N);
            CK(hipMemcpy(hc.data(), C, sizeof(bf16) * hc.size(), hipMemcpyDeviceToHost)); CK(hipMemcpy(hz.data(), Z, sizeof(bf16) * hz.size(), hipMemcpyDeviceToHost));
            CK(hipMemcpy(hr.data(), R, sizeof(float) * hr.size(), hipMemcpyDeviceToHost));
            long long bad = 0;
            for (size_t i = 0; i < hc.size(); ++i) {
                const float z = hr[i];
                float want = z;
                if (tc.epi) {
                    if ((float)hz[i] != (float)(bf16)z) { if (bad < 5) printf("z mismatch at %zu: %f vs %f\n", i, (float)hz[i], z); ++bad; }
                    want = (float)(0.5 * z * (1.0 + erf(z / sqrt(2.0))));
                    if (fabsf((float)hc[i] - want) > 0.01f * fabsf(want) + 1e-3f) { if (bad < 5) printf("y mismatch at %zu: %f vs %f\n", i, (float)hc[i], want); ++bad; }
                } else if ((float)hc[i] != (float)(bf16)want) { if (bad < 5) printf("mismatch at %zu: %f vs %f\n", i, (float)hc[i], want); ++bad; }
            }
            printf("exactness %dx%dx%d BM %d grid %d epi %d: %lld mismatches\n", tc.M, tc.N, tc.K, tc.bm, tc.grid, tc.epi, bad);
            if (bad) return 1;
            CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(Z)); CK(hipFree(R)); CK(hipFree(bias)); CK(hipFree(rows));
        }
    }
    if (quick) return 0;
    // ---- timing.  Operands rotate over NBUF copies so that A and the outputs of a launch come cold from HBM as in the training step
    // (the conv layers' activations are far larger than the 256 MB Infinity Cache; the encoder's are smaller: NBUF copies > 256 MB too).
    const Problem probs[] = {
        {"conv1 fwd", 8 * 24999, 512, 1536, 1024, 24999, 49999 * 512},
        {"conv2 fwd", 8 * 12499, 512, 1536, 1024, 12499, 24999 * 512},

        {"QKV", 6248, 2304, 768, 768, 0, 0},
        {"QKV half", 3124, 2304, 768, 768, 0, 0},
        {"FFN1", 6248, 3072, 768, 768, 0, 0},
        {"FFN1 half", 3124, 3072, 768, 768, 0, 0},
        {"4096x4096x8192", 4096, 4096, 8192, 8192, 0, 0},
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Problem& p : probs) {
        const size_t na = p.rpb ? (size_t)(p.M / p.rpb) * p.bs : (size_t)p.M * p.lda;
        const size_t nc = (size_t)p.M * p.N;
        const int NBUF = (int)std::max<size_t>(2, std::min<size_t>(8, (600ull << 20) / (na * 2) + 1));
        std::vector<bf16*> As(NBUF), Cs(NBUF), Zs(NBUF);
        bf16* B; CK(hipMalloc(&B, sizeof(bf16) * (size_t)p.N * p.K));
        hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, B, (long long)p.N * p.K, 6u, 0);
        for (int i = 0; i < NBUF; ++i) {
            CK(hipMalloc(&As[i], sizeof(bf16) * (na + 4096))); CK(hipMalloc(&Cs[i], sizeof(bf16) * nc)); CK(hipMalloc(&Zs[i], sizeof(bf16) * nc));
            hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, As[i], (long long)na + 4096, 5u + i, 0);
        }
        auto timeit = [&](const char* what, auto&& fn) {
            for (int i = 0; i < 2; ++i) fn(i % NBUF);
            float best = 1e30f, sum = 0;
            const int it = 3 * NBUF, rounds = 3;
            for (int r = 0; r < rounds; ++r) {
                CK(hipEventRecord(e0));
                for (int i = 0; i < it; ++i) fn(i % NBUF);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms / it); sum += ms / it;
            }
            printf("  %-40s best %7.1f us  mean %7.1f us  %5.0f TFLOP/s\n", what, best * 1e3, sum / rounds * 1e3, 2.0 * p.M * p.N * p.K / (best * 1e-3) / 1e12);
        };
        int rounds = 0;
        const int bm = best_bm(p.M, p.rpb ? (int)(p.M / p.rpb) : 1, p.N / BN, &rounds);
        printf("%s  M %d N %d K %d (%d cold copies)  best BM %d (%d rounds)\n", p.name, p.M, p.N, p.K, NBUF, bm, rounds);
        int stag = 0;
        auto args = [&](int i) { return V2Args{As[i], p.lda, p.rpb, p.bs, B, p.K, Cs[i], Zs[i], p.N, nullptr, p.M, p.N, p.K, 0, 0, 0, 0, 0, stag}; };
        timeit("p256 (round 3), plain", [&](int i) { launch_p256<0>(args(i)); });
        timeit("p256 (round 3), gelu + z", [&](int i) { launch_p256<1>(args(i)); });
        timeit("v2 BM 256, plain", [&](int i) { launch_v2<0>(args(i), 256, 0); });
        timeit("v2 BM 256, gelu + z", [&](int i) { launch_v2<1>(args(i), 256, 0); });
        char nm[64];
        snprintf(nm, sizeof nm, "v2 BM %d, plain", bm);
        timeit(nm, [&](int i) { launch_v2<0>(args(i), bm, 0); });
        snprintf(nm, sizeof nm, "v2 BM %d, gelu + z", bm);
        timeit(nm, [&](int i) { launch_v2<1>(args(i), bm, 0); });
        snprintf(nm, sizeof nm, "v2 BM %d, gelu, y only", bm);
        timeit(nm, [&](int i) { launch_v2<2>(args(i), bm, 0); });
        snprintf(nm, sizeof nm, "v2 BM %d, no gelu, y + z", bm);
        timeit(nm, [&](int i) { launch_v2<3>(args(i), bm, 0); });
        snprintf(nm, sizeof nm, "v2 BM %d, no gelu, y twice (same rows)", bm);
        timeit(nm, [&](int i) { launch_v2<4>(args(i), bm, 0); });
        for (int st : {100, 200, 400, 800}) {
            stag = st;
            snprintf(nm, sizeof nm, "v2 BM %d, gelu + z, stagger %d ns x (b/8 & 3)", bm, st * 10);
            timeit(nm, [&](int i) { launch_v2<1>(args(i), bm, 0); });
        }
        stag = 0;

        for (int i = 0; i < NBUF; ++i) { CK(hipFree(As[i])); CK(hipFree(Cs[i])); CK(hipFree(Zs[i])); }
        CK(hipFree(B));
    }
    return 0;
}
