// a8hip "stream" GEMM: bf16 NT products with many rows and N % 256 == 0 (the strided conv layers 1.. of the feature extractor,
// audio8/wav2vec2.py:424-429, as row-remapped GEMMs; any large Dense forward) on BM x 256 tiles, one persistent block per CU.
//
// Why a second NT kernel: a CU takes operands in at ~70-80 GB/s through L1 into LDS whatever the loop looks like,
// so FLOP/s per CU = that rate x tile FLOPs per staged byte: the 128 x 128 tiles of gemm.hip cap near 1.15 PFLOP/s, 256-wide
// tiles at twice that.  What makes the wide tile pay in THIS step (the round-2/3 256 x 256 kernel did not):
//   * tile height BM = 32 MI in {160, 192, 224, 256}, chosen per launch so that the tiles fill whole rounds of the 256 CUs
//     (conv layer 1: 1564 tiles of 256 rows = 6.1 rounds -> 7; 1786 tiles of 224 rows = 6.98 rounds);
//   * persistent blocks and ONE continuous k-tile stream across a block's tiles: the A operand (activations: cold, from HBM) is
//     requested three k-tiles ahead into a 3-slot ring, B (weights: L2 resident) two ahead into a 2-slot ring (160 KB of LDS);
//     the next tile's first k-tiles are in flight while the current tile finishes and stores (K % 384 == 0; otherwise a 2-slot A
//     ring, K % 128 == 0);
//   * the pipelined quadrant schedule (fragment reads of the next quadrant issued ahead of the MFMAs of the current one, one
//     barrier per k-tile placed mid-tile, compile-time ring slots: every LDS address is one of two lane constants + an immediate);
//   * every DMA wait is "all but my 4 youngest pieces" -- a count that never depends on the epilogue's stores: one extra wait in
//     front of the epilogue retires the next tile's first operands, the first k-tile of a tile then needs only the barrier;
//   * every tile has BM valid rows (the last tile of a batch item is shifted up to end with it: a few rows are computed twice,
//     identically), so sources are [tile base + k offset] (scalar) + [lane row * stride] (one register) and no store is predicated;
//   * epilogue without LDS: two v_permlane16_swap per fragment pair turn the accumulator layout's 8-byte column groups into
//     16-byte stores (16 rows x 64 contiguous bytes per wave-instruction).
// Results are bit-identical to gemm.hip's kernels (same MFMA, same k order, same epilogue arithmetic): tests/test_gemm_gpu.py.
// Measured standalone with cold operands (tools/ubench/gemm_v2.hip): conv layer 1 forward (GELU + pre-activation copy) 340 us
// against 419 (round-3 256 x 256 kernel) and ~430 (128 x 128 tiles); the second output stream, not the GELU, is most of the epilogue cost.
#include "a8_common.h"
#include "a8hip_internal.h"
#include <type_traits>

namespace {

constexpr int BN = 256, BK = 64, KROW = 128;
constexpr int TA = 256 * KROW, TB = BN * KROW;                  // 32 KB per ring slot

struct StreamArgs {
    const char* A; const char* B; char* C; char* Z; const char* Zin; const float* bias;
    int lda, a_bs;                 // elements: row stride of A, distance between batch items
    int ldb;
    int ldc, c_bs, ldz, z_bs;
    int K, rpb, tpb, tiles_m, tiles_n;      // rows per batch item, tiles per batch item
    int act;                       // A8_ACT_NONE / A8_ACT_GELU (Z = pre-activation copy out) / A8_ACT_MUL_GELU_GRAD, A8_ACT_ADD (Zin; rows as Z's)
};

struct TileAt { unsigned a, b; int bi, r0, n0; };       // byte offsets of the tile's first A row / B row; batch item, first row in it, first column

A8_DEV void glds16(const char* src, char* lds_dst_wave_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst_wave_uniform, 16, 0, 0);
}
A8_DEV unsigned pack2(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v = {(bf16)a, (bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

// MI = 16-row fragments per wave (5..8): tile height BM = 32 MI, wave row wr owns rows [16 MI wr, 16 MI (wr + 1)), wave column wc
// 64 of the 256 columns.  NA = slots of the A ring (3: K % 384 == 0; 2: K % 128 == 0).
// ZIN: the epilogues with a second input (x gelu'(zin), + zin) are their own instantiations: their zin registers and code stay out of
// the forward kernels' allocation.
template <int MI, int NA, bool ZIN>
__global__ __launch_bounds__(512, 2) void gemm_stream_kernel(const StreamArgs g) {
    static_assert(NA == 2 || NA == 3, "A ring depth");
    constexpr int BM = 32 * MI, WR = 16 * MI;
    constexpr int MI1 = MI - 4;                               // fragments in the second A quarter-read (the first always has 4)
    __shared__ __attribute__((aligned(1024))) char lds[NA * TA + 2 * TB];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6), wr = w >> 2, wc = w & 3;
    const int total = g.tiles_m * g.tiles_n, nk = g.K / BK, G = gridDim.x;
    constexpr int GROUP_M = 8;
    // tile v -> batch item, first row inside it (a tile never straddles two batch items), first column; scalar arithmetic only.
    // Past the end of the block's sequence its last tile is described again: the stream keeps issuing (valid, unused) pieces, so
    // every wave issues the same number of pieces at every position and one counted wait serves the whole kernel.
    auto tile_at = [&](int v) {
        TileAt d;
        const int tile = xcd_remap(min(v, total - 1 - ((total - 1 - (int)blockIdx.x) % G)), total);
        const int tpg = GROUP_M * g.tiles_n, grp = tile / tpg, first_m = grp * GROUP_M;
        const int gsz = min(GROUP_M, g.tiles_m - first_m), tin = tile - grp * tpg;
        const int mt = first_m + tin % gsz;
        d.bi = mt / g.tpb;
        d.r0 = min((mt - d.bi * g.tpb) * BM, g.rpb - BM);     // the last tile of a batch item ends with it (rows of its neighbour recomputed)
        d.n0 = (tin / gsz) * BN;
        d.a = ((unsigned)d.bi * (unsigned)g.a_bs + (unsigned)d.r0 * (unsigned)g.lda) * 2u;       // (spans < 4 GB: checked on the host)
        d.b = (unsigned)d.n0 * (unsigned)g.ldb * 2u;
        return d;
    };

    // ---- the DMA cursors.  32 one-KB pieces (8 rows) per operand per k-tile, 4 per wave (rows 64 apart); XOR swizzle on the source.
    // A piece's source = [tile base + 64 j rows + k offset] (scalar) + [lane's row * row stride + swizzled chunk] (one register);
    // pieces below the tile (wave-uniform) re-read piece 0.  LDS map: B slots at 0 / 32 KB, A slots from 64 KB.
    const int srow = 8 * w + (lane >> 3);
    const unsigned sch16 = (unsigned)(((lane & 7) ^ (srow & 7)) << 4);
    const unsigned arow = (unsigned)(srow * g.lda * 2) + sch16;
    const unsigned brow = (unsigned)(srow * g.ldb * 2) + sch16;
    const unsigned ajstep = 64u * g.lda * 2u, bjstep = 64u * g.ldb * 2u;
    TileAt cur = tile_at(blockIdx.x), nxt = tile_at(blockIdx.x + G);
    const char* a_base = g.A + cur.a; int a_kt = 0;
    const char* b_base = g.B + cur.b; int b_kt = 0;
    auto issue_a = [&](int slot) {
        char* ta = lds + 2 * TB + slot * TA;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const char* sj = a_base + ((8 * w + 64 * j < BM) ? j * ajstep : 0);       // wave-uniform
            glds16(sj + arow, ta + (w + 8 * j) * 1024);
        }
        a_base += KROW;
        if (++a_kt == nk) { a_kt = 0; a_base = g.A + nxt.a; }
    };
    auto issue_b = [&](int slot) {
        char* tb = lds + slot * TB;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16((b_base + j * bjstep) + brow, tb + (w + 8 * j) * 1024);
        b_base += KROW;
        if (++b_kt == nk) { b_kt = 0; b_base = g.B + nxt.b; }
    };

    f32x4 acc[MI][4];
    bf16x8 a0[2][4], a1[2][MI1], b0[2][2], b1[2][2];
    const int ra = wr * WR, cb = wc * 64;
    // Fragment reads.  A and B tiles share one image ([row][128 B], 16-byte chunk c of row r stored at c ^ (r & 7)), so a lane's
    // byte offset inside ANY 16-row fragment is one constant; the two k-steps differ by XOR 64.  Two lane constants in all (with the
    // wave's first row / column folded in); a slot's base is formed next to its reads by an opaque add (so that the compiler cannot
    // hoist five slots x two k-steps x two operands of loop-invariant addresses out of the loop and spill them -- scratch reloads
    // count in vmcnt and would drain the DMA ring), everything else is the instruction's immediate offset.
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

    // One stream position with A in ring slot SA, B in slot SB (compile-time).  On entry a0 / bx hold its first A / B quarters, by is
    // free; on exit a0 / by hold the first quarters of the next position (read while the last quadrant's MFMAs run): the two B
    // register sets swap roles every position.  Mid-position, behind the barrier: every fragment of this position is in registers,
    // so its slots are refilled -- B for position + 2, A for position + NA.
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
        // all but my 4 youngest pieces have landed (NA = 3: the A k-tile issued last stays in flight)
        if (wait) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA == 3 ? 4 : 0) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue_b(SB);
        issue_a(SA);
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

    // ---- prologue: positions 0, 1 (A and B) and, with the 3-slot ring, 2 (A)
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
        bf16* const crow0 = reinterpret_cast<bf16*>(g.C) + ((long long)done.bi * g.c_bs + (long long)done.r0 * g.ldc);
        bf16* const zrow0 = reinterpret_cast<bf16*>(g.Z) + ((long long)done.bi * g.z_bs + (long long)done.r0 * g.ldz);
        const bool gelu = !ZIN && g.act == A8_ACT_GELU, zout = !ZIN && g.Z != nullptr;
        const bool zmul = ZIN && g.act == A8_ACT_MUL_GELU_GRAD;
        // lanes of 16-lane row gq: even gq keeps fragment 2jj (its own 4 columns + the next row's), odd gq fragment 2jj+1
        const int col0 = done.n0 + cb + 16 * (gq & 1) + 8 * (gq >> 1);
        // epilogues with a second input (x gelu'(zin): conv data gradients; + zin): the lane's 16 bytes of every row fragment of a
        // column pair are requested ahead of that pair's arithmetic, in the layout of the stores; the same two swaps that build a
        // store turn a load back into the accumulator layout (the swap is an involution)
        u32x4 zv[ZIN ? MI : 1];
        const bf16* const zin0 = ZIN ? reinterpret_cast<const bf16*>(g.Zin) + ((long long)done.bi * g.z_bs + (long long)done.r0 * g.ldz) : nullptr;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            float bz[2][4];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (g.bias) bv = *reinterpret_cast<const f32x4*>(g.bias + done.n0 + cb + 32 * jj + 16 * q + 4 * gq);
#pragma unroll
                for (int e = 0; e < 4; ++e) bz[q][e] = bv[e];
            }
            const int col = col0 + 32 * jj;
            if constexpr (ZIN) {                         // the first column pair's zin; the second pair's is requested row fragment by row
                if (jj == 0) {                           // fragment as the first pair's registers free up (below): its latency is hidden
#pragma unroll
                    for (int i = 0; i < MI; ++i)
                        zv[i] = *reinterpret_cast<const u32x4*>(zin0 + ((unsigned)(ra + 16 * i + li) * (unsigned)g.ldz + (unsigned)col));
                }
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int rr = ra + 16 * i + li;
                unsigned zl[2], zh[2], yl[2], yh[2];
                if constexpr (ZIN) {
                    auto t0 = __builtin_amdgcn_permlane16_swap(zv[i][0], zv[i][2], false, false);
                    auto t1 = __builtin_amdgcn_permlane16_swap(zv[i][1], zv[i][3], false, false);
                    zl[0] = t0[0]; zl[1] = t0[1]; zh[0] = t1[0]; zh[1] = t1[1];
                    if (jj == 0) zv[i] = *reinterpret_cast<const u32x4*>(zin0 + ((unsigned)(ra + 16 * i + li) * (unsigned)g.ldz + (unsigned)(col + 32)));
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float x[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = acc[i][2 * jj + q][e] + bz[q][e];
                    if constexpr (ZIN) {
                        const float zf[4] = {__uint_as_float(zl[q] << 16), __uint_as_float(zl[q] & 0xFFFF0000u),
                                             __uint_as_float(zh[q] << 16), __uint_as_float(zh[q] & 0xFFFF0000u)};
                        if (zmul) mul_gelu_grad_n(x, zf);
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) x[e] += zf[e];
                        }
                    } else if (zout) { zl[q] = pack2(x[0], x[1]); zh[q] = pack2(x[2], x[3]); }
                    if (gelu) gelu_n(x);
                    yl[q] = pack2(x[0], x[1]); yh[q] = pack2(x[2], x[3]);
                }
                {
                    auto s0 = __builtin_amdgcn_permlane16_swap(yl[0], yl[1], false, false);
                    auto s1 = __builtin_amdgcn_permlane16_swap(yh[0], yh[1], false, false);
                    const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                    *reinterpret_cast<u32x4*>(crow0 + ((unsigned)rr * (unsigned)g.ldc + (unsigned)col)) = o;
                }
                if (zout) {
                    auto s0 = __builtin_amdgcn_permlane16_swap(zl[0], zl[1], false, false);
                    auto s1 = __builtin_amdgcn_permlane16_swap(zh[0], zh[1], false, false);
                    const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                    *reinterpret_cast<u32x4*>(zrow0 + ((unsigned)rr * (unsigned)g.ldz + (unsigned)col)) = o;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// the tile height that minimises rounds x (time per tile): MFMA time grows with BM, the operand feed with BM + 256.
// 256 rows (MI = 8) is not offered: at the 256-register cap those instantiations spill (2..16 registers, tests/test_codeobj_cpu.py),
// the reloads count in vmcnt and drain the DMA ring, and 224 rows measured faster on every layer anyway (conv layer 1 forward alone:
// 340 vs 350 us); a8_set_tuning key 7 = 256 still forces it (bit-identity tests).
int pick_bm(int batches, int rpb, int tiles_n, int cus) {
    int best = 0; double bc = 1e30;
    for (int bm = 160; bm <= 224; bm += 32) {
        if (rpb < bm) continue;
        const long long tiles = (long long)batches * cdiv(rpb, bm) * tiles_n;
        const long long rounds = (tiles + cus - 1) / cus;
        const double cost = (double)rounds * (bm + 160.0);
        if (cost < bc) { bc = cost; best = bm; }
    }
    return best;
}

}  // namespace

// Fills the kernel arguments if the problem fits the stream kernel.  a8_set_tuning key 5 = minimum M (0 = never); key 7 = forced
// tile height (0 = chosen per launch); key 19 = 1: plain (not row-remapped) problems too.
static bool stream_plan(const a8_gemm_desc* d, StreamArgs& g, int& bm, int& grid, bool& deep) {
    const int min_m = g_a8_tuning[5];
    if (min_m <= 0 || d->dtype != A8_BF16 || d->layout != A8_GEMM_NT || d->M < min_m) return false;
    // Opt-in by problem kind, not by size alone: only row-remapped A operands (the conv layers as GEMMs) take this kernel.  A plain Dense
    // product with M >= min_m (the encoder's projections at a large batch) stays on the 128 x 128 tiles: a persistent block owns its CU,
    // so nothing of the half-batch or weight-gradient streams runs beside it (+0.36 .. +0.47 ms per step measured).
    // a8_set_tuning key 19 = 1 lifts the restriction (tests, A/B runs).
    if (!d->a_rpb && g_a8_tuning[19] == 0) return false;
    if (d->N % 256 || d->K % 128 || d->K < 256 || d->out_f32 || d->accumulate || d->splitk > 1 || d->colsum || d->bias_rows || d->alpha != 1.0f) return false;
    const bool zin = d->act == A8_ACT_MUL_GELU_GRAD || d->act == A8_ACT_ADD;
    if (!(d->act == A8_ACT_NONE || d->act == A8_ACT_GELU || zin) || (zin && (!d->zin || d->zout)) || (!zin && d->zin) || d->b_rpb) return false;
    const int rpb = d->a_rpb ? (int)d->a_rpb : d->M;
    if (d->M % rpb) return false;
    const int batches = d->M / rpb;
    const bool hasz = d->zout || zin;
    if ((d->c_rpb && d->c_rpb != rpb) || (hasz && d->z_rpb && d->z_rpb != rpb)) return false;
    if (d->ldc % 8 || (hasz && d->ldz % 8) || ((uintptr_t)d->C % 16) || (d->zout && ((uintptr_t)d->zout % 16)) || (zin && ((uintptr_t)d->zin % 16)) ||
        (d->bias && ((uintptr_t)d->bias % 16)))
        return false;
    const long long a_bs = d->a_rpb ? d->a_bs : (long long)rpb * d->lda;
    const long long c_bs = d->c_rpb ? d->c_bs : (long long)rpb * d->ldc;
    const long long z_bs = hasz ? (d->z_rpb ? d->z_bs : (long long)rpb * d->ldz) : 0;
    // 32-bit byte offsets into A and B, 31-bit element offsets into C / Z
    const long long a_span = ((long long)(batches - 1) * a_bs + (long long)(rpb - 1) * d->lda + d->K) * 2;
    const long long b_span = ((long long)(d->N - 1) * d->ldb + d->K) * 2;
    const long long c_span = (long long)(batches - 1) * c_bs + (long long)(rpb - 1) * d->ldc + d->N;
    const long long z_span = hasz ? (long long)(batches - 1) * z_bs + (long long)(rpb - 1) * d->ldz + d->N : 0;
    if (a_span >= (1ll << 32) || b_span >= (1ll << 32) || c_span >= (1ll << 31) || z_span >= (1ll << 31) || a_bs >= (1ll << 31)) return false;
    static int cus = 0;
    if (!cus) {
        int dev = 0; hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return false;
        cus = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }
    bm = g_a8_tuning[7];
    if (!(bm == 160 || bm == 192 || bm == 224 || bm == 256) || rpb < bm) bm = pick_bm(batches, rpb, d->N / 256, cus);
    if (!bm) return false;
    // Aliasing rule (include/a8hip.h): the last tile of a batch item is shifted up to end with it, so when rpb % bm != 0 its top rows
    // are computed and stored by TWO tiles.  That is only sound while nothing a tile reads (A, B, zin) overlaps what another tile
    // writes (C, zout): an in-place "+ zin" / "x gelu'(zin)" (C == zin) would apply the second input twice on those rows.  Such a
    // problem goes to the 128 x 128 kernels, where every element is read and written once by the same lane.
    if (rpb % bm) {
        auto overlaps = [](const void* p, long long pbytes, const void* q, long long qbytes) {
            const uintptr_t a = (uintptr_t)p, b = (uintptr_t)q;
            return p && q && a < b + (uintptr_t)qbytes && b < a + (uintptr_t)pbytes;
        };
        const long long cb = c_span * 2, zb = z_span * 2;
        const void* outs[2] = {d->C, d->zout};
        const long long outb[2] = {cb, zb};
        for (int o = 0; o < 2; ++o) {
            if (!outs[o]) continue;
            if (overlaps(outs[o], outb[o], d->A, a_span) || overlaps(outs[o], outb[o], d->B, b_span) ||
                (zin && overlaps(outs[o], outb[o], d->zin, zb)))
                return false;
        }
    }
    g.A = (const char*)d->A; g.B = (const char*)d->B; g.C = (char*)d->C; g.Z = (char*)d->zout; g.Zin = (const char*)d->zin; g.bias = d->bias;
    g.lda = (int)d->lda; g.a_bs = (int)a_bs; g.ldb = (int)d->ldb;
    g.ldc = (int)d->ldc; g.c_bs = (int)c_bs; g.ldz = (int)d->ldz; g.z_bs = (int)z_bs;
    g.K = d->K; g.rpb = rpb; g.tpb = cdiv(rpb, bm); g.tiles_m = batches * g.tpb; g.tiles_n = d->N / 256;
    g.act = d->act;
    const int total = g.tiles_m * g.tiles_n;
    // blocks per CU's worth of grid (a8_set_tuning key 8; default 1: every block resident from the start, one static share each).
    // > 1: shorter static shares, the surplus blocks are dispatched as CUs free up -- the hardware balances the load when another
    // kernel (RCCL, the side stream's weight gradients) holds some CUs at launch, for one more ring prologue per block.
    const int over = g_a8_tuning[8] > 1 ? g_a8_tuning[8] : 1;
    grid = total < cus * over ? total : cus * over;
    deep = (d->K / 64) % 6 == 0;
    return true;
}

bool a8i_gemm_stream_eligible(const a8_gemm_desc* d) {
    StreamArgs g; int bm, grid; bool deep;
    return stream_plan(d, g, bm, grid, deep);
}

// Takes the problem if it fits the stream kernel (returns 1 and launches, 0 = not taken, < 0 = launch error code negated).
int a8i_gemm_stream_try(const a8_gemm_desc* d, void* stream) {
    StreamArgs g; int bm, grid; bool deep;
    if (!stream_plan(d, g, bm, grid, deep)) return 0;
    hipStream_t s = (hipStream_t)stream;
    const bool zin = g.Zin != nullptr;
#define A8_STREAM_LAUNCH(MI_) do { \
        if (deep && !zin) hipLaunchKernelGGL((gemm_stream_kernel<MI_, 3, false>), dim3(grid), dim3(512), 0, s, g); \
        else if (!zin) hipLaunchKernelGGL((gemm_stream_kernel<MI_, 2, false>), dim3(grid), dim3(512), 0, s, g); \
        else if (deep) hipLaunchKernelGGL((gemm_stream_kernel<MI_, 3, true>), dim3(grid), dim3(512), 0, s, g); \
        else hipLaunchKernelGGL((gemm_stream_kernel<MI_, 2, true>), dim3(grid), dim3(512), 0, s, g); } while (0)
    switch (bm) {
        case 160: A8_STREAM_LAUNCH(5); break;
        case 192: A8_STREAM_LAUNCH(6); break;
        case 224: A8_STREAM_LAUNCH(7); break;
        default: A8_STREAM_LAUNCH(8); break;
    }
#undef A8_STREAM_LAUNCH
    const int e = a8_check_launch("a8_gemm (stream)");
    return e ? -e : 1;
}
