// a8hip "wide" GEMM: bf16 NT products with N % 256 == 0 on 128 x 256 tiles that still fit TWO blocks per CU (round 5).
//
// Why: the 128 x 128 kernels of gemm.hip are bound by what a CU can take in through L1 into LDS (~60-80 GB/s per CU whatever the loop
// looks like: DESIGN section 5, profiles/r05_dma_feed_ubench.txt), i.e. by staged bytes per FLOP, and every tile with fewer bytes per FLOP
// built so far (gemm_stream.hip's BM x 256, the round-2/3 256 x 256 kernels) owns its CU (160 KB of LDS or 256 registers): on the
// encoder's products that costs more than it gains, because the half-batch and weight-gradient streams live on blocks of DIFFERENT
// kernels sharing a CU.  This kernel keeps the co-residency contract of gemm_a3_kernel -- 8 waves, <= 128 VGPRs, 80 KB of LDS -- and
// still stages 25 % fewer bytes per FLOP:
//   * tile 128 x 256, 8 waves as 2 (M) x 4 (N), each 64 x 64 (0.5 KB of LDS fragment reads per MFMA instead of 0.75);
//   * k-tile = 32 elements = 64-byte rows, ONE MFMA k-step per k-tile: a stage is 8 KB (A) + 16 KB (B); the ring holds 4 A stages and
//     3 B stages = 80 KB, A requested three k-tiles ahead, B two (56 KB in flight per block, 48 in gemm_a3_kernel);
//   * one barrier per k-tile with 16 MFMAs per wave between two barriers -- the same MFMA work per barrier as the 128 x 128 kernel's
//     64-deep k-tile, so the barrier count per FLOP does not grow;
//   * every wait is "all but my 4 youngest pieces" (B of the next k-tile and A of the next two may still be in flight);
//   * M tails by clamped source rows (a tile's surplus rows re-read row M - 1 and are never stored), so no lane predicate in the loop;
//   * LDS-free epilogue (gemm.hip's gemm_epilogue_direct at 64 rows per wave): bias / GELU + pre-activation copy / x gelu'(zin) / + zin,
//     optional column sums of the outputs (the b_1 gradient riding on the FFN data gradient) by DPP row sums + one atomic per column per wave.
// Results are bit-identical to gemm.hip's kernels (same MFMA, same k order within a row, same epilogue arithmetic): tests/test_gemm_gpu.py.
// Taken by a8_gemm for N >= a8_set_tuning key 24 (default 2304: QKV, FFN1 and the FFN2 data gradient; 0 = never).
#include "a8_common.h"
#include "a8hip_internal.h"

namespace {

constexpr int BM = 128, BN = 256, BK = 32, KROW = 64;
constexpr int TA = BM * KROW, TB = BN * KROW;                    // 8 KB / 16 KB per stage
constexpr int NA = 4, NB = 3;                                    // ring depths: A three k-tiles ahead, B two

struct WideArgs {
    const char* A; const char* B; char* C; char* Z; const char* Zin; const float* bias; float* colsum;
    int M, N, K, lda, ldb, ldc, ldz, tiles_m, tiles_n, act;
    float alpha;
};

A8_DEV void glds16(const char* src, char* lds_dst_wave_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst_wave_uniform, 16, 0, 0);
}
A8_DEV unsigned pack2(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 v = {(bf16)a, (bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
A8_DEV int swz(int row) { return (row >> 1) & 3; }              // 64-byte rows: 16-byte chunk c of row r stored at c ^ swz(r) (b128 reads conflict-free)

// row sums over the 16 lanes of a DPP row (lanes with equal lane >> 4): every lane of the row receives the sum
A8_DEV float row16_sum(float v) {
    v += dpp_f<0xB1>(v);          // quad_perm:[1,0,3,2]
    v += dpp_f<0x4E>(v);          // quad_perm:[2,3,0,1]
    v += dpp_f<0x141>(v);         // row_half_mirror
    v += dpp_f<0x140>(v);         // row_mirror
    return v;
}

__global__ __launch_bounds__(512, 4) void gemm_wide_kernel(const WideArgs g) {
    __shared__ __attribute__((aligned(1024))) char lds[NA * TA + NB * TB];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6), wm = w >> 2, wn = w & 3;
    const int li = lane & 15, gq = lane >> 4;
    // grouped tile order (gemm.hip): consecutive ids = one XCD's co-resident blocks cover 8 m-tiles x several n-tiles
    const int total = g.tiles_m * g.tiles_n;
    const int tile = xcd_remap(blockIdx.x, total);
    constexpr int GROUP_M = 8;
    const int tpg = GROUP_M * g.tiles_n, grp = tile / tpg, first_m = grp * GROUP_M;
    const int gsz = min(GROUP_M, g.tiles_m - first_m), tin = tile - grp * tpg;
    const int m0 = (first_m + tin % gsz) * BM, n0 = (tin / gsz) * BN;
    const int nk = g.K / BK;

    // ---- LDS-DMA sources: a 1 KB piece = 16 rows x 64 B (lane -> row l / 4, LDS slot l % 4 holding chunk slot ^ swz(row)).
    // A: 8 pieces per stage, one per wave; B: 16 pieces, two per wave.  Rows beyond M re-read row M - 1 (never stored).
    const int prow = lane >> 2;
    const char* a_src;
    const char* b_src[2];
    {
        const int rl = 16 * w + prow;
        const int row = min(m0 + rl, g.M - 1);
        a_src = g.A + ((long long)row * g.lda) * 2 + (((lane & 3) ^ swz(rl)) << 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rb = 16 * (w + 8 * j) + prow;
            b_src[j] = g.B + ((long long)(n0 + rb) * g.ldb) * 2 + (((lane & 3) ^ swz(rb)) << 4);
        }
    }
    int a_kt = 0, b_kt = 0;                       // k-tiles issued so far (scalar): past the last one the cursor stays (valid, unused bytes)
    auto issue_a = [&](int slot) {
        glds16(a_src, lds + slot * TA + w * 1024);
        if (++a_kt < nk) a_src += KROW;
    };
    auto issue_b = [&](int slot) {
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(b_src[j], lds + NA * TA + slot * TB + (w + 8 * j) * 1024);
        if (++b_kt < nk) { b_src[0] += KROW; b_src[1] += KROW; }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment reads: lane's byte offset inside any 16-row fragment of a stage (the wave's first row / column folded in by the caller)
    typedef const __attribute__((address_space(3))) bf16x8* LdsFrag;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned pat = lds0 + (unsigned)(li * KROW + ((gq ^ swz(li)) << 4));
    const unsigned pat_a = pat + (unsigned)(wm * 64 * KROW), pat_b = pat + (unsigned)(NA * TA + wn * 64 * KROW);

    // prologue: A k-tiles 0, 1, 2 and B k-tiles 0, 1 (issue order B then A inside a position, as in the loop)
    issue_b(0); issue_a(0);
    issue_b(1); issue_a(1);
    issue_a(2);
    int sa = 0, sb = 0;                           // ring slots of the k-tile being consumed
    for (int kt = 0; kt < nk; ++kt) {
        // all but my 4 youngest pieces have landed: B(kt + 1) (2 pieces) and A(kt + 1), A(kt + 2) may still be in flight ...
        // (issue order per iteration: B(kt + 2), then A(kt + 3); the prologue leaves exactly the same tail)
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ... and every wave has read k-tile kt - 1 out of the slots refilled now
        issue_b(sb == 0 ? 2 : sb - 1);            // B(kt + 2) -> slot (sb + 2) % 3
        issue_a(sa == 0 ? 3 : sa - 1);            // A(kt + 3) -> slot (sa + 3) % 4
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 fa[4], fb[4];
        const unsigned pa = pat_a + (unsigned)(sa * TA), pb = pat_b + (unsigned)(sb * TB);
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *(LdsFrag)(size_t)(pa + 16 * i * KROW);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *(LdsFrag)(size_t)(pb + 16 * j * KROW);
        __builtin_amdgcn_sched_barrier(0);
        // swapped operands: accumulator register r <-> 4 consecutive n, lane & 15 <-> m
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        sa = sa == NA - 1 ? 0 : sa + 1;
        sb = sb == NB - 1 ? 0 : sb + 1;
    }

    // ---- epilogue without LDS (gemm.hip's gemm_epilogue_direct at 64 rows per wave; the arithmetic in gemm_epilogue's order)
    {
#pragma clang fp contract(off)
        const bool zmul = g.act == A8_ACT_MUL_GELU_GRAD, zadd = g.act == A8_ACT_ADD, gelu = g.act == A8_ACT_GELU;
        const bool has_z = zmul || zadd;
        bf16* const C = reinterpret_cast<bf16*>(g.C);
        bf16* const Z = reinterpret_cast<bf16*>(g.Z);
        const bf16* const Zin = reinterpret_cast<const bf16*>(g.Zin);
        // the bias in the accumulator layout, 4 floats per (jj, q): requested here, not ahead of the main loop -- at the 128-register cap
        // its 16 registers spilled there (a co-resident block covers this one L2 round trip per tile)
        f32x4 bpre[4];
#pragma unroll
        for (int jq = 0; jq < 4; ++jq) {
            bpre[jq] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (g.bias) bpre[jq] = *reinterpret_cast<const f32x4*>(g.bias + n0 + wn * 64 + 16 * jq + 4 * gq);
        }
        bool rok[4];
        unsigned rowc[4], rowz[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + 16 * i + li;
            rok[i] = m < g.M;
            rowc[i] = (unsigned)(rok[i] ? m : 0) * (unsigned)g.ldc;           // (element offsets < 2^31: checked on the host)
            rowz[i] = (unsigned)(rok[i] ? m : 0) * (unsigned)g.ldz;
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int col = n0 + wn * 64 + 32 * jj + 16 * (gq & 1) + 8 * (gq >> 1);      // first of the lane's 8 stored columns
            u32x4 zv[4];
            if (has_z) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    zv[i] = (u32x4){0u, 0u, 0u, 0u};
                    if (rok[i]) zv[i] = *reinterpret_cast<const u32x4*>(Zin + rowz[i] + col);
                }
            }
            float cs[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned zl[2] = {0u, 0u}, zh[2] = {0u, 0u}, yl[2], yh[2];
                if (has_z) {
                    auto t0 = __builtin_amdgcn_permlane16_swap(zv[i][0], zv[i][2], false, false);
                    auto t1 = __builtin_amdgcn_permlane16_swap(zv[i][1], zv[i][3], false, false);
                    zl[0] = t0[0]; zl[1] = t0[1]; zh[0] = t1[0]; zh[1] = t1[1];
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float x[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { x[e] = acc[i][2 * jj + q][e] * g.alpha; x[e] += bpre[2 * jj + q][e] + 0.f; }
                    if (has_z) {
                        const float zf[4] = {__uint_as_float(zl[q] << 16), __uint_as_float(zl[q] & 0xFFFF0000u),
                                             __uint_as_float(zh[q] << 16), __uint_as_float(zh[q] & 0xFFFF0000u)};
                        if (zmul) mul_gelu_grad_n(x, zf);
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) x[e] += zf[e];
                        }
                    } else if (Z) { zl[q] = pack2(x[0], x[1]); zh[q] = pack2(x[2], x[3]); }
                    if (gelu) gelu_n(x);
                    yl[q] = pack2(x[0], x[1]); yh[q] = pack2(x[2], x[3]);
                    if (g.colsum) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) cs[q][e] += rok[i] ? x[e] : 0.f;
                    }
                }
                {
                    auto s0 = __builtin_amdgcn_permlane16_swap(yl[0], yl[1], false, false);
                    auto s1 = __builtin_amdgcn_permlane16_swap(yh[0], yh[1], false, false);
                    const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                    if (rok[i]) *reinterpret_cast<u32x4*>(C + rowc[i] + col) = o;
                }
                if (Z && !has_z) {
                    auto s0 = __builtin_amdgcn_permlane16_swap(zl[0], zl[1], false, false);
                    auto s1 = __builtin_amdgcn_permlane16_swap(zh[0], zh[1], false, false);
                    const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                    if (rok[i]) *reinterpret_cast<u32x4*>(Z + rowz[i] + col) = o;
                }
            }
            if (g.colsum) {                     // column sums of the wave's 64 rows: the 16 lanes of a DPP row hold the same 4 columns per (jj, q)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float s = row16_sum(cs[q][e]);
                        if (li == 0) atomicAdd(g.colsum + n0 + wn * 64 + 32 * jj + 16 * q + 4 * gq + e, s);
                    }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the ring's look-ahead pieces past the last k-tile)
}

// host: does the problem fit?  (plain row-major operands, no remaps: the encoder's and the heads' Dense products)
bool wide_plan(const a8_gemm_desc* d, WideArgs& g) {
    const int min_n = g_a8_tuning[24];
    if (min_n <= 0 || d->dtype != A8_BF16 || d->layout != A8_GEMM_NT || d->N < min_n) return false;
    if (d->N % BN || d->K % BK || d->K < 3 * BK || d->M < 1 || d->out_f32 || d->accumulate || d->splitk > 1 || d->bias_rows) return false;
    if (d->a_rpb || d->b_rpb || d->c_rpb || d->z_rpb) return false;
    const bool zin = d->act == A8_ACT_MUL_GELU_GRAD || d->act == A8_ACT_ADD;
    if (!(d->act == A8_ACT_NONE || d->act == A8_ACT_GELU || zin)) return false;
    if ((zin && (!d->zin || d->zout || d->bias)) || (!zin && d->zin)) return false;
    const bool hasz = d->zout || zin;
    if (d->lda % 8 || d->ldb % 8 || d->ldc % 8 || (hasz && d->ldz % 8)) return false;
    if (((uintptr_t)d->A % 16) || ((uintptr_t)d->B % 16) || ((uintptr_t)d->C % 16) || (d->zout && ((uintptr_t)d->zout % 16)) ||
        (zin && ((uintptr_t)d->zin % 16)) || (d->bias && ((uintptr_t)d->bias % 16)))
        return false;
    if ((long long)d->M * d->ldc >= (1ll << 31) || (hasz && (long long)d->M * d->ldz >= (1ll << 31))) return false;
    g.A = (const char*)d->A; g.B = (const char*)d->B; g.C = (char*)d->C; g.Z = (char*)d->zout; g.Zin = (const char*)d->zin;
    g.bias = d->bias; g.colsum = d->colsum;
    g.M = d->M; g.N = d->N; g.K = d->K;
    g.lda = (int)d->lda; g.ldb = (int)d->ldb; g.ldc = (int)d->ldc; g.ldz = (int)d->ldz;
    g.tiles_m = cdiv(d->M, BM); g.tiles_n = d->N / BN;
    g.act = d->act; g.alpha = d->alpha;
    return (long long)g.tiles_m * g.tiles_n < (1ll << 30);
}

}  // namespace

// Takes the problem if it fits (returns 1 and launches, 0 = not taken, < 0 = launch error code negated).
int a8i_gemm_wide_try(const a8_gemm_desc* d, void* stream) {
    WideArgs g;
    if (!wide_plan(d, g)) return 0;
    hipLaunchKernelGGL(gemm_wide_kernel, dim3(g.tiles_m * g.tiles_n), dim3(512), 0, (hipStream_t)stream, g);
    const int e = a8_check_launch("a8_gemm (wide)");
    return e ? -e : 1;
}
