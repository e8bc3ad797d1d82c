// a8hip: multi-head self-attention core, forward + backward, flash style (scores never reach HBM).
// Replaces eight_mile SeqScaledDotProductAttention as configured at audio8/wav2vec2.py:613-622
// (scores = q k^T / sqrt(d_k); masked_fill(mask == 0, -1e9); softmax; dropout; . v), head dim 64.
//
// MI355X design notes
//  * The head slices are read in place from the [B*T, ld] projection outputs (no head transposes).
//  * Scores are computed TRANSPOSED (S^T = K Q^T) so that the accumulator of a 16x16 tile --
//    lane l, reg r = S^T[key 4(l>>4)+r][query l&15] -- is already, register for register, the MFMA
//    operand of the following product (P V, dS K, P^T dO, dS^T Q): the k-slot order of an MFMA is free
//    as long as both operands agree, so the other operand is read from LDS in the accumulator's key order
//    with ds_read_b64_tr_b16 (bf16) / ds_read_b32 (f32).  P and dS never touch LDS.
//  * Softmax statistics are lane-local per query (query = lane&15) + two xor-shuffles over the 4 key groups.
//  * One LDS image per 64x64 tile serves both the row reads (b128, XOR-swizzled) and the transposed reads.
//  * Dropout regenerates its keep-mask from a counter hash of (b,h,query,key) in all three kernels: one 32-bit
//    word per QUAD of adjacent keys, h = fin((quad_index + seed) * C1), an 8-bit field per key (the drop probability
//    is realised in 1/256 steps and the 1/(1-p) scale uses the realised value).  The quad index is affine in the tile /
//    fragment coordinates, so the product is carried as a running sum (one add per word): the forward and dQ kernels,
//    whose lanes hold 4 adjacent keys of one query per accumulator, hash once per accumulator.
//  * The softmax inner loops are VALU-bound at head dim 64, so they are written for instruction count: scores stay
//    raw, exp2(fma(s, scale, -max)) with the bare v_exp_f32; the running max is only raised when a tile exceeds it
//    by more than 2^8 (probabilities stay <= 256, exact same result after the final 1/l); the constant factors
//    1/(1-p) and the softmax scale are applied once to the accumulators; tails need no per-element tests in the
//    backward kernels because staged rows beyond T are zero.
//  * Backward is two kernels (dK/dV stationary per key block; dQ stationary per query block): no atomics,
//    bit-reproducible.  delta = rowsum(dO*O) is a tiny pre-pass.
#include "a8_common.h"
#include "a8hip_internal.h"

namespace {

constexpr int DH = 64;      // head dim
constexpr int BT = 64;      // tile rows (queries or keys)

// ------------------------------------------------------------------ 64x64 LDS tile image
// row_frag:  MFMA operand of 16 tile rows (k-slots = head-dim columns), read along the rows (b128, XOR-swizzled).
// xt_frag:   MFMA operand of 16 tile COLUMNS c0.. (k-slots = 32 tile rows kb.. in ACCUMULATOR order: slot 4g+r of
//            the first 16 rows, then of the next 16), read transposed -- it pairs with a probability fragment built
//            from two score accumulators (pfrag), so P / dS go from accumulator registers straight into the next MFMA.
template <typename T> struct ATile;
template <> struct ATile<bf16> {
    static constexpr int BYTES = BT * 128;
    static constexpr int CPR = 8;                       // 16-byte chunks per row
    typedef bf16x8 RF;
    typedef bf16x8 XT;
    typedef bf16x8 PF;
    static A8_DEV int off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
    static A8_DEV RF row_frag(const char* t, int rb, int ks, int lane) {
        const int row = rb + (lane & 15);
        return *reinterpret_cast<const bf16x8*>(t + off(row, ks * 4 + (lane >> 4)));
    }
    static A8_DEV XT xt_frag(const char* t, int kb, int c0, int lane) {
        const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
        const int r0 = kb + 4 * g + q, r1 = r0 + 16;
        const int ch = (c0 >> 3) + (p >> 1), sub = (p & 1) << 3;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, t + off(r0, ch) + sub));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, t + off(r1, ch) + sub));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        s16x8 xv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, xv);
    }
    static A8_DEV PF pfrag(f32x4 p0, f32x4 p1) {
        return (bf16x8){(bf16)p0[0], (bf16)p0[1], (bf16)p0[2], (bf16)p0[3], (bf16)p1[0], (bf16)p1[1], (bf16)p1[2], (bf16)p1[3]};
    }
    static A8_DEV f32x4 mma_xp(XT x, PF p, f32x4 acc) { return Mma<bf16>::mma(x, p, acc); }
};
template <> struct ATile<float> {
    static constexpr int LDF = 68;                      // floats per row (272 B, 16-byte aligned)
    static constexpr int BYTES = BT * LDF * 4;
    static constexpr int CPR = 16;
    typedef float RF;
    struct XT { float a[8]; };
    struct PF { f32x4 p0, p1; };
    static A8_DEV int off(int row, int chunk) { return (row * LDF + chunk * 4) * 4; }
    static A8_DEV RF row_frag(const char* t, int rb, int ks, int lane) {
        return *reinterpret_cast<const float*>(t + ((rb + (lane & 15)) * LDF + ks * 4 + (lane >> 4)) * 4);
    }
    static A8_DEV XT xt_frag(const char* t, int kb, int c0, int lane) {
        const int g = lane >> 4, i = lane & 15;
        XT x;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            x.a[r] = *reinterpret_cast<const float*>(t + ((kb + 4 * g + r) * LDF + c0 + i) * 4);
            x.a[4 + r] = *reinterpret_cast<const float*>(t + ((kb + 16 + 4 * g + r) * LDF + c0 + i) * 4);
        }
        return x;
    }
    static A8_DEV PF pfrag(f32x4 p0, f32x4 p1) { return PF{p0, p1}; }
    static A8_DEV f32x4 mma_xp(const XT& x, const PF& p, f32x4 acc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = Mma<float>::mma(x.a[r], p.p0[r], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = Mma<float>::mma(x.a[4 + r], p.p1[r], acc);
        return acc;
    }
};

template <typename T, int NT> struct AStage {       // global -> registers -> LDS for one 64x64 tile, NT threads
    static constexpr int CPR = ATile<T>::CPR;
    static constexpr int NCH = BT * CPR / NT;        // chunks per thread
    static constexpr int RSTEP = NT / CPR;
    u32x4 r[NCH];
    A8_DEV void load(const T* base, long long ld, int row0, int nrows) {   // base -> element (row 0, col 0 of the head)
        const int c = threadIdx.x % CPR, r0 = threadIdx.x / CPR;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int row = row0 + r0 + j * RSTEP;
            r[j] = row < nrows ? ld16(base + (long long)row * ld + c * (16 / (int)sizeof(T))) : (u32x4){0u, 0u, 0u, 0u};
        }
    }
    A8_DEV void store(char* tile) const {
        const int c = threadIdx.x % CPR, r0 = threadIdx.x / CPR;
#pragma unroll
        for (int j = 0; j < NCH; ++j) st16(tile + ATile<T>::off(r0 + j * RSTEP, c), r[j]);
    }
};

// fragments of one row held in registers straight from global memory (natural k-slot order)
template <typename T> struct RowFrags;
template <> struct RowFrags<bf16> {
    bf16x8 f[2];
    A8_DEV void load(const bf16* rowp, bool valid, int lane) {
        const int g = lane >> 4;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            f[ks] = valid ? *reinterpret_cast<const bf16x8*>(rowp + ks * 32 + 8 * g) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
};
template <> struct RowFrags<float> {
    float f[16];
    A8_DEV void load(const float* rowp, bool valid, int lane) {
        const int g = lane >> 4;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) f[ks] = valid ? rowp[ks * 4 + g] : 0.f;
    }
};

// sum over the head dimension of a[row][:] * b[row][:] for the row this lane holds (lanes li, li+16, li+32, li+48 share a row)
A8_DEV float row_dot(const RowFrags<bf16>& a, const RowFrags<bf16>& b) {
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 8; ++i) s = __builtin_fmaf((float)a.f[ks][i], (float)b.f[ks][i], s);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    return s;
}
A8_DEV float row_dot(const RowFrags<float>& a, const RowFrags<float>& b) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s = __builtin_fmaf(a.f[i], b.f[i], s);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    return s;
}

// acc[j][i] (i = 0..3) = 16x16 blocks of Tile . rows_j^T for G register-resident row groups: each tile fragment is
// read from LDS once and feeds G MFMAs.   acc[j][i][r] = sum_d Tile[16i+4g+r][d] * rows_j[lane&15][d]
template <typename T, int G> A8_DEV void tile_times_rows(const char* tile, const RowFrags<T> (&rf)[G], f32x4 (&acc)[G][4], int lane) {
    constexpr int NKS = DH / Mma<T>::KI;
#pragma unroll
    for (int j = 0; j < G; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const typename ATile<T>::RF a = ATile<T>::row_frag(tile, 16 * i, ks, lane);
#pragma unroll
            for (int j = 0; j < G; ++j) acc[j][i] = Mma<T>::mma(a, rf[j].f[ks], acc[j][i]);
        }
}
// out[j][db] += (tile columns 16db..)^T-contracted with the probability fragments pf[j][half] (half = tile rows 32*half..)
template <typename T, int G> A8_DEV void xt_times_p(const char* tile, const typename ATile<T>::PF (&pf)[G][2], f32x4 (&out)[G][4], int lane) {
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const typename ATile<T>::XT x = ATile<T>::xt_frag(tile, 32 * hf, 16 * db, lane);
#pragma unroll
            for (int j = 0; j < G; ++j) out[j][db] = ATile<T>::mma_xp(x, pf[j][hf], out[j][db]);
        }
}

// Streaming two 64x64 tiles (K,V or Q,dO) through LDS.  16-bit: two LDS buffers, the tile after next is in flight in
// registers, ONE barrier per tile.  f32 (parity mode, 2 x 17 KB tiles): single buffer, two barriers.
template <typename T> constexpr int attn_nbuf() { return sizeof(T) == 2 ? 2 : 1; }

struct AttnP {
    const char *q, *k, *v, *o, *dout;
    char *out, *dq, *dk, *dv;
    float *lse, *delta;
    float* dbias;                  // optional f32 [3][H*64] += column sums of dq | dk | dv (the q/k/v bias gradients)
    const uint8_t* keymask;
    int B, T, H;
    long long ld_qkv, ld_o, ld_d;
    float scale, pscale, dscale;   // scale = softmax scale * log2(e) (scores and lse live in the log2 domain); dscale = softmax scale
    float ipscale;                 // 1 / pscale = the realised keep probability (exact: (256 - thr) / 256)
    float lazy;                    // raise the running max only past this margin (log2 units)
    uint32_t thr, seed;            // thr: 8-bit keep threshold (0 = dropout off)
    int Th;                        // key quads per query row = ceil(T / 4)
    int bh0;                       // (batch, head) index of this call's first pair in the LOGICAL batch the dropout counters
                                   // belong to (a call on utterances [b0, b0 + B) of a larger batch passes b0 * H: same mask as the whole-batch call)
};

constexpr uint32_t HC1 = 0x9E3779B1u, HC2 = 0x2C1B3C6Du;
constexpr float MASKED_LOG2 = -1.4426950408889634e9f;      // masked_fill(-1e9) in the log2 domain
A8_DEV float mask_fill_value(int nattended) { return nattended == 0 ? 0.f : MASKED_LOG2; }   // (see mask_prefix_len)
// dropout word of key quad `idx`: fin((idx + seed) * HC1); callers carry the product as a running sum
A8_DEV uint32_t hfin(uint32_t h) { h ^= h >> 15; return h * HC2; }
A8_DEV uint32_t pair_base(int bh, int T, int Th, int q) { return ((uint32_t)bh * (uint32_t)T + (uint32_t)q) * (uint32_t)Th; }
A8_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// lane (4k + R)'s value in all four lanes of quad k (v_mov_b32_dpp quad_perm:[R,R,R,R]; the compiler folds it into the consumer)
template <int R> A8_DEV uint32_t quad_bcast(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, R * 0x55, 0xF, 0xF, true); }

// 1-D grid -> (tile block, batch*head).  Workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so
// when B*H divides by 8 every block of one (batch, head) is placed on the same XCD and its K/V (or Q/dO) panel
// is fetched from HBM/MALL once per XCD instead of once per block.
A8_DEV void block_coords(int nblk, int nbh, int& blk, int& bh) {
    const int L = blockIdx.x;
    if (nbh % 8 == 0) {
        const int xcd = L & 7, slot = L >> 3;
        bh = xcd + 8 * (slot / nblk);
        blk = slot % nblk;
    } else {
        bh = L / nblk;
        blk = L % nblk;
    }
}

// Key-padding masks are, in practice, PREFIX masks: eight_mile's sequence_mask marks the first n_b frames of utterance b (audio8/train.py:33,
// audio8/wav2vec2.py:703-708).  One pass over the utterance's mask row (T bytes, all loads independent) tells a wave whether its row is
// such a mask with at least one attended key, and n_b: the kernel then IS the unmasked kernel over n_b keys -- key tiles beyond n_b are
// never staged, only the boundary tile tests keys (exact: a masked score of -1e9 has probability exp(-1e9 - max) = 0 in f32 whenever
// one key is attended).  Returns -1 for a mask with holes and 0 when NOTHING is attended: the per-key path.  Round 5 tested km[key] per
// score in every tile: the masked forward / dQ kernels ran 80 / 88 us per layer at cfg5 against 37 / 49 unmasked at cfg2 (profiles/r06_cfg5_*).
// An utterance without any attended key: the reference's softmax over T equal scores of -1e9 is uniform, and its gradient reaches V only
// (masked_fill cuts q and k off).  -1e9 + log2(T) is not representable in f32, so such a row stands in 0 for the fill value (mask_fill_value:
// the softmax is shift-invariant) and masked keys get dS = 0 explicitly in the per-key path.
struct MaskScan {
    int v[16];
    float cnt, last1, first0;
    A8_DEV void acc(int k0, int lane) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float k = (float)(k0 + lane + 64 * i);
            if (v[i] > 0) { cnt += 1.f; last1 = fmaxf(last1, k); }
            else if (v[i] == 0) first0 = fminf(first0, k);
        }
    }
    A8_DEV void fetch(const uint8_t* __restrict__ km, int T, int k0, int lane) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int k = k0 + lane + 64 * i; v[i] = k < T ? (int)km[k] : -1; }
    }
    // requests the row's first 1024 bytes and returns with the loads in flight (rows longer than that: the rest first, synchronously)
    A8_DEV void load(const uint8_t* __restrict__ km, int T, int lane) {
        cnt = 0.f; last1 = -1.f; first0 = 16777216.f;
        for (int k0 = 1024; k0 < T; k0 += 1024) { fetch(km, T, k0, lane); acc(k0, lane); }
        fetch(km, T, 0, lane);
    }
    A8_DEV int finish(int lane) {
        acc(0, lane);
        const float c = wave_sum(cnt), l1 = wave_max(last1), f0 = -wave_max(-first0);
        const int n = c > 0.f ? (f0 > l1 ? (int)c : -1) : 0;
        return __builtin_amdgcn_readfirstlane(n);
    }
};
A8_DEV int mask_prefix_len(const uint8_t* __restrict__ km, int T, int lane) {
    MaskScan ms;
    ms.load(km, T, lane);
    return ms.finish(lane);
}

// out[c] += scale * sum over the block's valid rows of acc[.][c]: acc[j][db][r] is column 16 db + 4 (lane >> 4) + r of the row
// (lane & 15) of group j.  Rows reduce by shuffles inside each 16-lane group, waves through `red` ([NW][64] floats of free LDS),
// then one atomic per column per block.  Call with every thread of the block (it synchronises).
template <int G, int NW> A8_DEV void block_colsum(const f32x4 (&acc)[G][4], const bool (&valid)[G], float scale, float* __restrict__ out,
                                                  float* red, int wv, int lane) {
    __syncthreads();                                   // the tile buffers `red` aliases are no longer read
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < G; ++j) v += valid[j] ? acc[j][db][r] : 0.f;
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 4, 64);
            v += __shfl_xor(v, 8, 64);
            if ((lane & 15) == 0) red[wv * 64 + 16 * db + 4 * (lane >> 4) + r] = v;
        }
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < NW; ++q) t += red[q * 64 + threadIdx.x];
        atomicAdd(out + threadIdx.x, t * scale);
    }
}

template <typename T> A8_DEV void store4(T* p, float a, float b, float c, float d);
template <> A8_DEV void store4<bf16>(bf16* p, float a, float b, float c, float d) {
    *reinterpret_cast<bf16x4*>(p) = (bf16x4){(bf16)a, (bf16)b, (bf16)c, (bf16)d};
}
template <> A8_DEV void store4<float>(float* p, float a, float b, float c, float d) { *reinterpret_cast<f32x4*>(p) = (f32x4){a, b, c, d}; }

// Common prologue/stream loop for a pair of tiles (ta, tb) read from (pa, lda) / (pb, ldb).
//   bf16:  tile t is computed from buffer t&1 while tile t+1 is written to the other buffer and tile t+2 is loading.
#define ATTN_STREAM_BEGIN(pa, lda_, pb, ldb_)                                                              \
    AStage<T, NT> sa_, sb_;                                                                                \
    sa_.load(pa, lda_, 0, P.T);                                                                            \
    sb_.load(pb, ldb_, 0, P.T);                                                                            \
    if constexpr (NBUF == 2) {                                                                             \
        sa_.store(lds);                                                                                    \
        sb_.store(lds + TB);                                                                               \
        if (ntiles > 1) { sa_.load(pa, lda_, BT, P.T); sb_.load(pb, ldb_, BT, P.T); }                      \
        __syncthreads();                                                                                   \
    }
#define ATTN_STREAM_TILE(t, pa, lda_, pb, ldb_, ta, tb)                                                    \
    const char *ta, *tb;                                                                                   \
    if constexpr (NBUF == 2) {                                                                             \
        ta = lds + ((t) & 1) * 2 * TB;                                                                     \
        tb = ta + TB;                                                                                      \
        if ((t) + 1 < ntiles) {                                                                            \
            char* nx = lds + (((t) + 1) & 1) * 2 * TB;                                                     \
            sa_.store(nx);                                                                                 \
            sb_.store(nx + TB);                                                                            \
            if ((t) + 2 < ntiles) { sa_.load(pa, lda_, ((t) + 2) * BT, P.T); sb_.load(pb, ldb_, ((t) + 2) * BT, P.T); } \
        }                                                                                                  \
    } else {                                                                                               \
        ta = lds;                                                                                          \
        tb = lds + TB;                                                                                     \
        __syncthreads();                                                                                   \
        sa_.store(lds);                                                                                    \
        sb_.store(lds + TB);                                                                               \
        __syncthreads();                                                                                   \
        if ((t) + 1 < ntiles) { sa_.load(pa, lda_, ((t) + 1) * BT, P.T); sb_.load(pb, ldb_, ((t) + 1) * BT, P.T); } \
    }
#define ATTN_STREAM_END() if constexpr (NBUF == 2) __syncthreads();

// ================================================================== forward
// block = NW waves, each wave owns G groups of 16 queries (query = 16*group + lane&15)
template <typename T, int NW, int G, int MW = 1>
__global__ __launch_bounds__(NW * 64, MW) void attn_fwd_kernel(const AttnP P) {
    constexpr int NT = NW * 64, BQ = NW * G * 16, TB = ATile<T>::BYTES, NBUF = attn_nbuf<T>();
    __shared__ __attribute__((aligned(16))) char lds[NBUF * 2 * TB];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
    int blk, bh;
    block_coords((P.T + BQ - 1) / BQ, P.B * P.H, blk, bh);
    const int b = bh / P.H, h = bh % P.H;
    const T* qb = reinterpret_cast<const T*>(P.q) + (long long)b * P.T * P.ld_qkv + h * DH;
    const T* kb = reinterpret_cast<const T*>(P.k) + (long long)b * P.T * P.ld_qkv + h * DH;
    const T* vb = reinterpret_cast<const T*>(P.v) + (long long)b * P.T * P.ld_qkv + h * DH;
    const uint8_t* km = P.keymask ? P.keymask + (long long)b * P.T : nullptr;
    const int ntiles = (P.T + BT - 1) / BT;
    const float maskv = mask_fill_value(km ? mask_prefix_len(km, P.T, lane) : 1);

    RowFrags<T> qf[G];
    f32x4 oacc[G][4];
    float m[G], l[G];                                 // m: the max the probabilities are referred to (log2 domain)
    uint32_t hb[G];                                   // dropout word base of the current tile
    int qi[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
        qi[j] = blk * BQ + (wv * G + j) * 16 + li;
        qf[j].load(qb + (long long)qi[j] * P.ld_qkv, qi[j] < P.T, lane);
        m[j] = -INFINITY; l[j] = 0.f;
        hb[j] = (pair_base(bh + P.bh0, P.T, P.Th, qi[j]) + (uint32_t)g + P.seed) * HC1;
#pragma unroll
        for (int i = 0; i < 4; ++i) oacc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    ATTN_STREAM_BEGIN(kb, P.ld_qkv, vb, P.ld_qkv)
    for (int kt_i = 0; kt_i < ntiles; ++kt_i) {
        ATTN_STREAM_TILE(kt_i, kb, P.ld_qkv, vb, P.ld_qkv, kt, vt)
        f32x4 s[G][4];
        tile_times_rows<T, G>(kt, qf, s, lane);       // s[j][i][r] = raw S^T[key 16i+4g+r][query group j, li]
        const bool edge = km != nullptr || (kt_i + 1) * BT > P.T;      // wave-uniform: masks / tail only here
        typename ATile<T>::PF pf[G][2];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            float mt;
            if (!edge) {
                mt = fmaxf(fmaxf(s[j][0][0], s[j][0][1]), fmaxf(s[j][0][2], s[j][0][3]));
#pragma unroll
                for (int i = 1; i < 4; ++i) mt = fmaxf(fmaxf(mt, s[j][i][0]), fmaxf(fmaxf(s[j][i][1], s[j][i][2]), s[j][i][3]));
                mt *= P.scale;
            } else {
                const int key0 = kt_i * BT + 4 * g;
                mt = -INFINITY;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = key0 + 16 * i + r;
                        float x = s[j][i][r] * P.scale;
                        if (km && key < P.T && !km[key]) x = maskv;
                        if (key >= P.T) x = -INFINITY;
                        s[j][i][r] = x;              // already scaled on this path
                        mt = fmaxf(mt, x);
                    }
            }
            mt = rows_max(mt);
            if (__any(mt > m[j] + P.lazy)) {          // first tile: m = -inf -> always taken, alpha = 0
                const float mn = fmaxf(m[j], mt);
                const float alpha = fast_exp2(m[j] - mn);
                m[j] = mn;
                l[j] *= alpha;
#pragma unroll
                for (int db = 0; db < 4; ++db) oacc[j][db] *= alpha;
            }
            float ls = 0.f;
            if (!edge) {
                const float nb = -m[j];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { s[j][i][r] = fast_exp2(__builtin_fmaf(s[j][i][r], P.scale, nb)); ls += s[j][i][r]; }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { s[j][i][r] = fast_exp2(s[j][i][r] - m[j]); ls += s[j][i][r]; }
            }
            l[j] += ls;
            if (P.thr) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {        // keys 16i + 4g + r, r = 0..3: one quad, one word
                    const uint32_t h0 = hfin(hb[j] + (uint32_t)(4 * i) * HC1);
                    s[j][i][0] = (h0 & 0xFFu) >= P.thr ? s[j][i][0] : 0.f;
                    s[j][i][1] = ((h0 >> 8) & 0xFFu) >= P.thr ? s[j][i][1] : 0.f;
                    s[j][i][2] = ((h0 >> 16) & 0xFFu) >= P.thr ? s[j][i][2] : 0.f;
                    s[j][i][3] = (h0 >> 24) >= P.thr ? s[j][i][3] : 0.f;
                }
                hb[j] += (uint32_t)(BT / 4) * HC1;
            }
            pf[j][0] = ATile<T>::pfrag(s[j][0], s[j][1]);
            pf[j][1] = ATile<T>::pfrag(s[j][2], s[j][3]);
        }
        xt_times_p<T, G>(vt, pf, oacc, lane);
        ATTN_STREAM_END()
    }
#pragma unroll
    for (int j = 0; j < G; ++j) {
        const float lj = rows_sum(l[j]);
        if (qi[j] < P.T) {
            const float inv = P.pscale / lj;          // 1/(1-p) of the dropout is applied here, once
            T* op = reinterpret_cast<T*>(P.out) + ((long long)b * P.T + qi[j]) * P.ld_o + h * DH + 4 * g;
#pragma unroll
            for (int db = 0; db < 4; ++db) store4<T>(op + 16 * db, oacc[j][db][0] * inv, oacc[j][db][1] * inv, oacc[j][db][2] * inv, oacc[j][db][3] * inv);
            if (g == 0) P.lse[((long long)bh) * P.T + qi[j]] = m[j] + log2f(lj);     // log2-domain log-sum-exp
        }
    }
}

// Occupancy the backward kernels are compiled for.  Left to itself the register allocator aims at 3 waves per SIMD (a 168-VGPR budget),
// parks the dK / dV (dQ) accumulators in AGPRs and copies them in and out around every VGPR-form MFMA: 129 of the 398 vector
// instructions of the dK/dV loop were v_accvgpr_read / write -- and 164 + 40 registers run at 2 waves per SIMD anyway.
#define A8_ATTN_WPE(T, G, MW) ((MW) > 1 ? (MW) : ((G) == 1 && sizeof(T) == 2 ? 2 : 1))
// Dropout / key-mask presence as template flags (0 = off, 1 = on, 2 = read from the parameters at run time): with run-time tests the
// 16 scores of a lane are 16 basic blocks -- a wave-uniform branch each -- and nothing of one score's exp / hash chain overlaps the next's.
#define A8_ATTN_FLAG(F, runtime) ((F) == 2 ? (runtime) : (F) == 1)

// ================================================================== backward A: dK, dV (key block stationary)
// block = NW waves, each wave owns G groups of 16 keys; streams Q / dO tiles (+ lse, delta of the tile's queries)
template <typename T, int NW, int G, int MW = 1, int DROP = 2, int MASK = 2>
__global__ __launch_bounds__(NW * 64, MW) __attribute__((amdgpu_waves_per_eu(A8_ATTN_WPE(T, G, MW), A8_ATTN_WPE(T, G, MW))))
void attn_bwd_kv_kernel(const AttnP P) {
    constexpr int NT = NW * 64, BK = NW * G * 16, TB = ATile<T>::BYTES, NBUF = attn_nbuf<T>();
    const bool drop = A8_ATTN_FLAG(DROP, P.thr != 0), masked = A8_ATTN_FLAG(MASK, P.keymask != nullptr);
    __shared__ __attribute__((aligned(16))) char lds[NBUF * 2 * TB];
    __shared__ __attribute__((aligned(16))) float s_stat[NBUF][2][BT];     // [buffer][lse | delta][query in tile]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
    int blk, bh;
    block_coords((P.T + BK - 1) / BK, P.B * P.H, blk, bh);
    const int b = bh / P.H, h = bh % P.H;
    const T* qb = reinterpret_cast<const T*>(P.q) + (long long)b * P.T * P.ld_qkv + h * DH;
    const T* kb = reinterpret_cast<const T*>(P.k) + (long long)b * P.T * P.ld_qkv + h * DH;
    const T* vb = reinterpret_cast<const T*>(P.v) + (long long)b * P.T * P.ld_qkv + h * DH;
    const T* dob = reinterpret_cast<const T*>(P.dout) + (long long)b * P.T * P.ld_o + h * DH;
    const int ntiles = (P.T + BT - 1) / BT;
    const uint32_t ThC = (uint32_t)P.Th * HC1;

    RowFrags<T> kf[G], vf[G];
    f32x4 dk[G][4], dv[G][4];
    int key[G];
    bool key_masked[G];
    // Dropout word of (query qq, key): fin(hk + (qq - tile base - 4g) * ThC), 8-bit field (key & 3).  The four lanes of a quad hold
    // the four keys of one key quad, so they need the SAME 16 words per tile: lane (key & 3) = r hashes the words of the queries
    // 16 i + 4 g + r only (4 instead of 16 hashes; the 32-bit multiply is a quarter-rate instruction) and the quad exchanges them by DPP.
    uint32_t hk[G];
    const uint32_t fmask = 0xFFu << (8 * (li & 3)), fthr = P.thr << (8 * (li & 3));
#pragma unroll
    for (int j = 0; j < G; ++j) {
        key[j] = blk * BK + (wv * G + j) * 16 + li;
        const bool ok = key[j] < P.T;
        kf[j].load(kb + (long long)key[j] * P.ld_qkv, ok, lane);
        vf[j].load(vb + (long long)key[j] * P.ld_qkv, ok, lane);
        key_masked[j] = masked && ok && !P.keymask[(long long)b * P.T + key[j]];
        hk[j] = (pair_base(bh + P.bh0, P.T, P.Th, 4 * g + (li & 3)) + (uint32_t)(key[j] >> 2) + P.seed) * HC1;
#pragma unroll
        for (int i = 0; i < 4; ++i) { dk[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    }
    float maskv = MASKED_LOG2;
    if (masked) {
        // a prefix key mask (mask_prefix_len): a block whose keys all lie beyond the attended prefix has p = 0 for every score -- dK = dV = 0
        const int nv = mask_prefix_len(P.keymask + (long long)b * P.T, P.T, lane);
        maskv = mask_fill_value(nv);
        if (nv > 0 && blk * BK >= nv) {                // block-uniform
#pragma unroll
            for (int j = 0; j < G; ++j)
                if (key[j] < P.T) {
                    T* dkp = reinterpret_cast<T*>(P.dk) + ((long long)b * P.T + key[j]) * P.ld_d + h * DH + 4 * g;
                    T* dvp = reinterpret_cast<T*>(P.dv) + ((long long)b * P.T + key[j]) * P.ld_d + h * DH + 4 * g;
#pragma unroll
                    for (int db = 0; db < 4; ++db) { store4<T>(dkp + 16 * db, 0.f, 0.f, 0.f, 0.f); store4<T>(dvp + 16 * db, 0.f, 0.f, 0.f, 0.f); }
                }
            return;
        }
    }
    // per-tile softmax statistics ride along with the tiles: rows beyond T get Q = dO = 0 (staged zero) and
    // lse = delta = 0, which makes dS = 0 and P^T dO = 0 without any per-element test
    auto load_stat = [&](int t) -> float {
        const int qq = t * BT + (threadIdx.x & (BT - 1));
        if (threadIdx.x >= 2 * BT || qq >= P.T) return 0.f;
        return (threadIdx.x < BT ? P.lse : P.delta)[(long long)bh * P.T + qq];
    };
    auto store_stat = [&](int buf, float v) {
        if (threadIdx.x < 2 * BT) s_stat[buf][threadIdx.x >> 6][threadIdx.x & (BT - 1)] = v;
    };
    static_assert(NT >= 2 * BT, "statistics staging needs 128 threads");
    float st = load_stat(0);
    if constexpr (NBUF == 2) { store_stat(0, st); if (ntiles > 1) st = load_stat(1); }
    ATTN_STREAM_BEGIN(qb, P.ld_qkv, dob, P.ld_o)
    for (int qt_i = 0; qt_i < ntiles; ++qt_i) {
        const float* stat;
        if constexpr (NBUF == 2) {
            stat = &s_stat[qt_i & 1][0][0];
            if (qt_i + 1 < ntiles) { store_stat((qt_i + 1) & 1, st); if (qt_i + 2 < ntiles) st = load_stat(qt_i + 2); }
        } else {
            stat = &s_stat[0][0][0];
        }
        ATTN_STREAM_TILE(qt_i, qb, P.ld_qkv, dob, P.ld_o, qt, dot)
        if constexpr (NBUF == 1) {                     // single buffer: the tile macro has just passed its two barriers
            __syncthreads();
            store_stat(0, st);
            __syncthreads();
            if (qt_i + 1 < ntiles) st = load_stat(qt_i + 1);
        }
        f32x4 s[G][4], dp[G][4];
        tile_times_rows<T, G>(qt, kf, s, lane);        // s[j][i][r]  = raw S[q 16i+4g+r][key group j, li]
        tile_times_rows<T, G>(dot, vf, dp, lane);      // dp[j][i][r] = (dO V^T)[q][key]
        typename ATile<T>::PF pp[G][2], pd[G][2];
#pragma unroll
        for (int j = 0; j < G; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 lse4 = *reinterpret_cast<const f32x4*>(stat + 16 * i + 4 * g);
                const f32x4 del4 = *reinterpret_cast<const f32x4*>(stat + BT + 16 * i + 4 * g);      // delta / pscale (see the dQ kernel)
                uint32_t wr[4] = {0u, 0u, 0u, 0u};
                if (drop) {
                    const uint32_t w = hfin(hk[j] + (uint32_t)(16 * i) * ThC);       // this lane's share: query 16 i + 4 g + (key & 3)
                    wr[0] = quad_bcast<0>(w); wr[1] = quad_bcast<1>(w); wr[2] = quad_bcast<2>(w); wr[3] = quad_bcast<3>(w);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = __builtin_fmaf(s[j][i][r], P.scale, -lse4[r]);
                    if (masked && key_masked[j]) x = maskv - lse4[r];
                    const float p = fast_exp2(x);      // keys beyond T: harmless garbage in rows that are never stored
                    float pdrop = p;
                    if (drop) pdrop = (wr[r] & fmask) >= fthr ? p : 0.f;
                    s[j][i][r] = pdrop;                                                  // P (dropped, unscaled) for dV
                    dp[j][i][r] = __builtin_fmaf(pdrop, dp[j][i][r], -(p * del4[r]));    // dS / (softmax scale * pscale) for dK
                    if (masked && key_masked[j]) dp[j][i][r] = 0.f;                      // a filled score does not depend on q or k
                }
            }
            hk[j] += (uint32_t)BT * ThC;
            pp[j][0] = ATile<T>::pfrag(s[j][0], s[j][1]);
            pp[j][1] = ATile<T>::pfrag(s[j][2], s[j][3]);
            pd[j][0] = ATile<T>::pfrag(dp[j][0], dp[j][1]);
            pd[j][1] = ATile<T>::pfrag(dp[j][2], dp[j][3]);
        }
        xt_times_p<T, G>(dot, pp, dv, lane);
        xt_times_p<T, G>(qt, pd, dk, lane);
        ATTN_STREAM_END()
    }
    const float kscale = P.dscale * P.pscale;
#pragma unroll
    for (int j = 0; j < G; ++j)
        if (key[j] < P.T) {
            T* dkp = reinterpret_cast<T*>(P.dk) + ((long long)b * P.T + key[j]) * P.ld_d + h * DH + 4 * g;
            T* dvp = reinterpret_cast<T*>(P.dv) + ((long long)b * P.T + key[j]) * P.ld_d + h * DH + 4 * g;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                store4<T>(dkp + 16 * db, dk[j][db][0] * kscale, dk[j][db][1] * kscale, dk[j][db][2] * kscale, dk[j][db][3] * kscale);
                store4<T>(dvp + 16 * db, dv[j][db][0] * P.pscale, dv[j][db][1] * P.pscale, dv[j][db][2] * P.pscale, dv[j][db][3] * P.pscale);
            }
        }
    if (P.dbias) {                                     // block-uniform: k and v bias gradients ride on this kernel
        bool ok[G];
#pragma unroll
        for (int j = 0; j < G; ++j) ok[j] = key[j] < P.T;
        float* red = reinterpret_cast<float*>(lds);
        const int d = P.H * DH;
        block_colsum<G, NW>(dk, ok, kscale, P.dbias + d + h * DH, red, wv, lane);
        block_colsum<G, NW>(dv, ok, P.pscale, P.dbias + 2 * d + h * DH, red, wv, lane);
    }
}

// ================================================================== backward B: dQ (query block stationary)
template <typename T, int NW, int G, int MW = 1, int DROP = 2, int MASK = 2>
__global__ __launch_bounds__(NW * 64, MW) __attribute__((amdgpu_waves_per_eu(A8_ATTN_WPE(T, G, MW), A8_ATTN_WPE(T, G, MW))))
void attn_bwd_q_kernel(const AttnP P) {
    constexpr int NT = NW * 64, BQ = NW * G * 16, TB = ATile<T>::BYTES, NBUF = attn_nbuf<T>();
    const bool drop = A8_ATTN_FLAG(DROP, P.thr != 0), masked = A8_ATTN_FLAG(MASK, P.keymask != nullptr);
    __shared__ __attribute__((aligned(16))) char lds[NBUF * 2 * TB];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
    int blk, bh;
    block_coords((P.T + BQ - 1) / BQ, P.B * P.H, blk, bh);
    const int b = bh / P.H, h = bh % P.H;
    const T* qb = reinterpret_cast<const T*>(P.q) + (long long)b * P.T * P.ld_qkv + h * DH;
    const T* kb = reinterpret_cast<const T*>(P.k) + (long long)b * P.T * P.ld_qkv + h * DH;
    const T* vb = reinterpret_cast<const T*>(P.v) + (long long)b * P.T * P.ld_qkv + h * DH;
    const T* dob = reinterpret_cast<const T*>(P.dout) + (long long)b * P.T * P.ld_o + h * DH;
    const T* ob = reinterpret_cast<const T*>(P.o) + (long long)b * P.T * P.ld_o + h * DH;
    const uint8_t* km = masked ? P.keymask + (long long)b * P.T : nullptr;
    // prefix key mask: the unmasked kernel over the first Tk keys (tiles beyond them are never staged, only the boundary tile tests keys);
    // any other mask: km[key] per score (mask_prefix_len)
    int Tk = P.T;
    bool holes = false;
    float maskv = MASKED_LOG2;
    if (masked) {
        const int nv = mask_prefix_len(km, P.T, lane);
        if (nv > 0) Tk = nv; else holes = true;
        maskv = mask_fill_value(nv);
    }
    const int ntiles = (Tk + BT - 1) / BT;

    RowFrags<T> qf[G], dof[G];
    f32x4 dq[G][4];
    float nlse[G], ndelta[G];
    uint32_t hb[G];
    int qi[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
        qi[j] = blk * BQ + (wv * G + j) * 16 + li;
        const bool ok = qi[j] < P.T;
        qf[j].load(qb + (long long)qi[j] * P.ld_qkv, ok, lane);
        dof[j].load(dob + (long long)qi[j] * P.ld_o, ok, lane);
        nlse[j] = ok ? -P.lse[(long long)bh * P.T + qi[j]] : 0.f;
        {   // delta = rowsum(dO * O) of this block's own queries: computed here (this kernel runs first) and published for
            // the dK/dV kernel -- no separate pass over O and dO
            RowFrags<T> of;
            of.load(ob + (long long)qi[j] * P.ld_o, ok, lane);
            const float dl = row_dot(of, dof[j]);
            ndelta[j] = -dl;
            if (ok && g == 0) P.delta[(long long)bh * P.T + qi[j]] = dl * P.ipscale;      // (the dK/dV kernel works in units of pscale)
        }
        hb[j] = (pair_base(bh + P.bh0, P.T, P.Th, qi[j]) + (uint32_t)g + P.seed) * HC1;
#pragma unroll
        for (int i = 0; i < 4; ++i) dq[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    ATTN_STREAM_BEGIN(kb, P.ld_qkv, vb, P.ld_qkv)
    for (int kt_i = 0; kt_i < ntiles; ++kt_i) {
        ATTN_STREAM_TILE(kt_i, kb, P.ld_qkv, vb, P.ld_qkv, kt, vt)
        f32x4 s[G][4], dp[G][4];
        tile_times_rows<T, G>(kt, qf, s, lane);        // raw S^T[key][q]
        tile_times_rows<T, G>(vt, dof, dp, lane);      // (V dO^T)[key][q]
        // keys beyond T: staged K = V = 0, so dS is finite garbage that multiplies zero K rows -- no tail test needed
        typename ATile<T>::PF pf[G][2];
#pragma unroll
        for (int j = 0; j < G; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t h0 = 0;
                if (drop) h0 = hfin(hb[j] + (uint32_t)(4 * i) * HC1);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = __builtin_fmaf(s[j][i][r], P.scale, nlse[j]);
                    bool filled = false;                                        // (per-key path only) the score is the fill value: dS = 0
                    if (masked) {                                              // block-uniform tests
                        const int key = kt_i * BT + 4 * g + 16 * i + r;
                        if (holes) {
                            filled = key >= P.T || !km[min(key, P.T - 1)];     // (keys beyond T: with lse near -1e9 even a zero K row would give p = inf)
                            if (filled) x = maskv + nlse[j];
                        } else if ((kt_i + 1) * BT > Tk) { if (key >= Tk) x = MASKED_LOG2 + nlse[j]; }
                    }
                    const float p = fast_exp2(x);
                    float dpv = dp[j][i][r];
                    if (drop) dpv = (((h0 >> (8 * r)) & 0xFFu) >= P.thr) ? dpv : 0.f;
                    s[j][i][r] = p * __builtin_fmaf(dpv, P.pscale, ndelta[j]);      // dS^T[key][q] / softmax scale
                    if (masked && holes && filled) s[j][i][r] = 0.f;
                }
            }
            hb[j] += (uint32_t)(BT / 4) * HC1;
            pf[j][0] = ATile<T>::pfrag(s[j][0], s[j][1]);
            pf[j][1] = ATile<T>::pfrag(s[j][2], s[j][3]);
        }
        xt_times_p<T, G>(kt, pf, dq, lane);
        ATTN_STREAM_END()
    }
#pragma unroll
    for (int j = 0; j < G; ++j)
        if (qi[j] < P.T) {
            T* dqp = reinterpret_cast<T*>(P.dq) + ((long long)b * P.T + qi[j]) * P.ld_d + h * DH + 4 * g;
#pragma unroll
            for (int db = 0; db < 4; ++db) store4<T>(dqp + 16 * db, dq[j][db][0] * P.dscale, dq[j][db][1] * P.dscale, dq[j][db][2] * P.dscale, dq[j][db][3] * P.dscale);
        }
    if (P.dbias) {                                     // block-uniform: the q bias gradient rides on this kernel
        bool ok[G];
#pragma unroll
        for (int j = 0; j < G; ++j) ok[j] = qi[j] < P.T;
        block_colsum<G, NW>(dq, ok, P.dscale, P.dbias + h * DH, reinterpret_cast<float*>(lds), wv, lane);
    }
}

// ================================================================== forward, 16-bit: 32x32x16 MFMA, a wave owns 32 queries
// Round 3.  The 16x16 kernels above spend 12.7 VALU instructions per MFMA on the softmax (profiles/r02_attn_sq_counters.txt): the
// forward pass is bound by vector-instruction issue, not by the matrix pipe.  This kernel is built around instruction count:
//  * S^T = K Q^T on v_mfma_f32_32x32x16_bf16: lane (query = lane & 31, half = lane >> 5) holds 16 of a key block's 32 scores of ITS
//    query, so row statistics are lane-local (one permlane32 swap per kernel for the sum, one per rescale for the max), the score
//    registers 8s .. 8s+7 ARE the B operand of O^T += V^T P^T for 16-key step s (k order (j & 3) + 8 (j >> 2) + 4 half, matched by
//    the transposed V reads), and an MFMA blocks the vector issue for 8 of its 32 cycles instead of 8 of 16;
//  * no running-max pass: p = exp2(s * c - m) against the reference m of the last rescale, and the tile's own row sum (needed
//    anyway) doubles as the overflow test -- a lane whose 32 probabilities sum beyond 2^8 sends the wave through the rescale
//    path (true max, O and l rescaled, scores recomputed from the K tile still in LDS); the first tile (m = -inf) takes it by itself;
//  * K / V tiles arrive by LDS-DMA (global_load_lds, 16 B per lane) into a 3-deep ring: ONE barrier per 64-key tile, two tiles in
//    flight, no staging registers or LDS stores; K image: 128-byte rows, 16-byte chunk c at c ^ ((row >> 1) & 7) (conflict-free
//    b128 row reads of 32 rows); V image: 256-byte blocks [key quad][d half][4 keys x 32 d] -- a half-wave's transposed read
//    (ds_read_b64_tr_b16) is exactly one block, all 64 banks once, and every V address is one lane constant + an immediate;
//  * the dropout words are those of the kernels above (one 32-bit hash per key quad, an 8-bit field per key): backward
//    regenerates the identical mask.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int F32_TILE = 16384;                 // K image 8 KB | V image 8 KB
constexpr float F32_SUM_LIMIT = 256.0f;         // per-lane tile sum beyond this -> rescale path (every p <= 2^8 otherwise)


A8_DEV void attn_glds16(const char* src, char* lds_dst_wave_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst_wave_uniform, 16, 0, 0);
}
A8_DEV f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

template <int NW>
__global__ __launch_bounds__(NW * 64, 3) void attn_fwd32_kernel(const AttnP P) {
    constexpr int BQ = NW * 32, NBUF = 3;
    __shared__ __attribute__((aligned(1024))) char lds[NBUF * F32_TILE + 1024];      // + a dummy page for the DMAs issued past the last tile
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int blk, bh;
    block_coords((P.T + BQ - 1) / BQ, P.B * P.H, blk, bh);
    const int b = bh / P.H, h = bh % P.H;
    const bf16* qb = reinterpret_cast<const bf16*>(P.q) + (long long)b * P.T * P.ld_qkv + h * DH;
    const char* kb_ = reinterpret_cast<const char*>(reinterpret_cast<const bf16*>(P.k) + (long long)b * P.T * P.ld_qkv + h * DH);
    const char* vb_ = reinterpret_cast<const char*>(reinterpret_cast<const bf16*>(P.v) + (long long)b * P.T * P.ld_qkv + h * DH);
    const uint8_t* km = P.keymask ? P.keymask + (long long)b * P.T : nullptr;
    const int ntiles = (P.T + BT - 1) / BT;                  // tiles of the operands (DMA addressing / clamps)
    const long long row_bytes = P.ld_qkv * 2, tile_bytes = (long long)BT * row_bytes;
    // Key mask: the row scan's loads are the OLDEST vector loads of the wave (retired with the Q loads by the counted wait below).
    // Prefix mask -> Tk attended keys, ntl key tiles walked; anything else -> per-key tests (holes).
    MaskScan ms;
    if (km) ms.load(km, P.T, lane);

    // ---- this wave's four DMA pieces of a tile (1 KB each; 16 per tile: 0..7 = K rows 8p.., 8..15 = V blocks 4(p-8)..; with 5 waves
    // four pieces are staged twice -- same bytes, same place).  Rows beyond T are clamped to row T-1 (finite data whose
    // probabilities are zero): only the last tile has any, and it takes its own, precomputed, offsets.
    unsigned soff[4], soff_clamp[4]; int dst[4]; bool is_k[4];      // soff_clamp: what the last tile's row clamp takes off (a value select:
                                                                    // a select between two ARRAYS becomes a dynamic index and parks both in scratch)
    const int last_rows = P.T - (ntiles - 1) * BT;           // valid rows of the last tile (1..64)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pid = (wv + NW * j) & 15;
        int row, chunk;
        if (pid < 8) {                                   // K: lane -> row 8 pid + lane / 8, LDS slot lane % 8 holds chunk slot ^ key(row)
            row = 8 * pid + (lane >> 3);
            chunk = (lane & 7) ^ ((row >> 1) & 7);
        } else {                                         // V: lane -> block 4 (pid - 8) + lane / 16 = (key quad, d half); key lane/4 % 4, chunk lane % 4
            const int blkid = 4 * (pid - 8) + (lane >> 4);
            row = 4 * (blkid >> 1) + ((lane >> 2) & 3);
            chunk = 4 * (blkid & 1) + (lane & 3);
        }
        is_k[j] = pid < 8;
        soff[j] = (unsigned)(row * (int)row_bytes + chunk * 16);
        soff_clamp[j] = (unsigned)((row - min(row, last_rows - 1)) * (int)row_bytes);
        dst[j] = pid * 1024;
    }
    // (nlive is a parameter, not a capture: a by-reference capture that is assigned between two calls keeps the whole closure -- and the
    // offset arrays it points to -- in scratch memory)
    auto issue = [&](int t, int slot, int nlive) __attribute__((always_inline)) {
        const bool live = t < nlive;                     // wave-uniform
        const int tt = live ? t : ntiles - 1;            // past the end: a harmless re-read of the last tile into the dummy page
        const char* kbase = kb_ + (long long)tt * tile_bytes;
        const char* vbase = vb_ + (long long)tt * tile_bytes;
        char* base = live ? lds + slot * F32_TILE : lds + NBUF * F32_TILE;
        const bool lastt = tt == ntiles - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            attn_glds16((is_k[j] ? kbase : vbase) + (soff[j] - (lastt ? soff_clamp[j] : 0u)), base + (live ? dst[j] : 0));
    };

    // ---- Q fragments (B operand: lane holds Q[query][16 ks + 8 half + 0..7]); dropout word base; accumulators
    const int q = blk * BQ + wv * 32 + r;
    const bool q_ok = q < P.T;
    const bool wave_live = blk * BQ + wv * 32 < P.T;           // wave-uniform
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        qf[ks] = q_ok ? *reinterpret_cast<const bf16x8*>(qb + (long long)q * P.ld_qkv + 16 * ks + 8 * hh) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    issue(0, 0, ntiles);                         // (liveness by T here: a second tile beyond a short attended prefix is staged and never read)
    issue(1, 1, ntiles);
    // The Q loads (and the mask scan's) are older than the eight DMAs: retire them HERE, with a wait the compiler's counter pass can see
    // (an inline-asm wait is invisible to it) -- otherwise it protects the first use of qf inside the loop with a vmcnt(0) that would
    // drain the ring's prefetches in every iteration.
    __builtin_amdgcn_s_waitcnt(0x0F78);          // vmcnt(8) alone (expcnt 7, lgkmcnt 15)
    const int nv = km ? ms.finish(lane) : 1;
    const bool holes = km != nullptr && nv <= 0;
    const float maskv = mask_fill_value(nv);
    const int Tk = (km != nullptr && nv > 0) ? nv : P.T, ntl = (Tk + BT - 1) / BT;
    f32x16 o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
    float m = -INFINITY, l = 0.f;
    uint32_t hbase = (pair_base(bh + P.bh0, P.T, P.Th, q) + (uint32_t)hh + P.seed) * HC1;       // word of key quad 2 mq + half of key block 0, tile 0
    const int kkey = (r >> 1) & 7;                                                       // K row read: row 32 kb + r, chunk (2 ks + half) ^ kkey
    const int vlane = hh * 512 + ((lane & 15) >> 2) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
    int kaddr[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kaddr[ks] = r * 128 + (((2 * ks + hh) ^ kkey) << 4);

    int slot = 0;
    for (int t = 0; t < ntl; ++t) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");          // this wave's pieces of tile t have landed (tile t+1 may be in flight)
        __builtin_amdgcn_s_barrier();                             // ... and everybody else's; nobody reads tile t-1's slot any more
        int nslot = slot + 2; nslot = nslot >= NBUF ? nslot - NBUF : nslot;
        const char* kt = lds + slot * F32_TILE;
        const char* vt = kt + 8192;
        slot = slot + 1 == NBUF ? 0 : slot + 1;
        if (wave_live) {                                          // (a wave whose 32 queries all lie beyond T only stages and synchronises)
        f32x16 s[2], pp[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                s[kb] = mfma32(*reinterpret_cast<const bf16x8*>(kt + kb * 4096 + kaddr[ks]), qf[ks], s[kb]);
        }
        // ---- probabilities against the current reference max; the row sum is the overflow test
        const bool edge = holes || (t + 1) * BT > Tk;             // wave-uniform
        float tsum = 0.f;
        bool rescale = edge || t == 0;                            // (the first tile has no reference max yet: straight to the rescale path)
        if (!rescale) {
            const float nb = -m;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float p = fast_exp2(__builtin_fmaf(s[kb][i], P.scale, nb));
                    pp[kb][i] = p;
                    tsum += p;       // ONE dependent chain on purpose: independent partial sums get packed into v_pk_add_f32, which issues
                }                    // slower than two plain adds, and a wave issues at most one vector instruction per ~4 cycles anyway.
                                     // (No inline asm here: hipcc pads no transcendental -> VALU hazard for an asm consumer.)
            rescale = __any(!(tsum <= F32_SUM_LIMIT));            // (also true for inf / NaN)
        }
        if (rescale) {
            float mt = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float x = s[kb][i] * P.scale;
                    if (edge) {
                        const int key = t * BT + 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * hh;
                        if (holes && key < P.T && !km[key]) x = maskv;
                        if (key >= Tk) x = key < P.T ? MASKED_LOG2 : -INFINITY;
                    }
                    s[kb][i] = x;
                    mt = fmaxf(mt, x);
                }
            {   // the other half-wave holds the other 32 keys of the same query
                auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
                mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            const float mn = fmaxf(m, mt);                        // finite: every tile starts below T, masked keys are -1e9
            const float alpha = fast_exp2(m - mn);                // first tile: exp2(-inf) = 0
            m = mn;
            l *= alpha;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[db][i] *= alpha;
            tsum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float p = fast_exp2(s[kb][i] - m);
                    pp[kb][i] = p;
                    tsum += p;
                }
        }
        l += tsum;
        // ---- dropout (one hash word per key quad = 4 consecutive registers), bf16 packing, O^T += V^T P^T
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if (P.thr) {
#pragma unroll
                for (int mq = 0; mq < 4; ++mq) {
                    const uint32_t h0 = hfin(hbase + (uint32_t)(8 * kb + 2 * mq) * HC1);
                    pp[kb][4 * mq + 0] = (h0 & 0xFFu) >= P.thr ? pp[kb][4 * mq + 0] : 0.f;
                    pp[kb][4 * mq + 1] = ((h0 >> 8) & 0xFFu) >= P.thr ? pp[kb][4 * mq + 1] : 0.f;
                    pp[kb][4 * mq + 2] = ((h0 >> 16) & 0xFFu) >= P.thr ? pp[kb][4 * mq + 2] : 0.f;
                    pp[kb][4 * mq + 3] = (h0 >> 24) >= P.thr ? pp[kb][4 * mq + 3] : 0.f;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = (bf16x8){(bf16)pp[kb][8 * s2 + 0], (bf16)pp[kb][8 * s2 + 1], (bf16)pp[kb][8 * s2 + 2], (bf16)pp[kb][8 * s2 + 3],
                                           (bf16)pp[kb][8 * s2 + 4], (bf16)pp[kb][8 * s2 + 5], (bf16)pp[kb][8 * s2 + 6], (bf16)pp[kb][8 * s2 + 7]};
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const char* va = vt + vlane + kb * 4096 + s2 * 2048 + db * 256;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, va));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, va + 1024));
                    typedef short s16x8 __attribute__((ext_vector_type(8)));
                    const s16x8 xv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    o[db] = mfma32(__builtin_bit_cast(bf16x8, xv), pf, o[db]);
                }
            }
        }
        hbase += (uint32_t)(BT / 4) * HC1;
        }
        // Tile t+2 goes into the slot tile t-1 was read from (free since this iteration's barrier).  Issued at the END of the body:
        // the transposed-read builtin carries no alias information, so hipcc drains every pending LDS-DMA (vmcnt(0)) in front of the
        // first V read that follows one -- placed here, that wait falls in the NEXT iteration, a whole K phase + softmax after the issue.
        issue(t + 2, nslot, ntl);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the dummy DMAs past the last tile
    // ---- epilogue: the two half-waves of a query hold complementary halves of its row sum
    {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l), __float_as_uint(l), false, false);
        l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    // Registers 4 mq .. 4 mq + 3 hold head-dim columns 32 db + 8 mq + 4 half + 0..3 of the lane's query: 8 bytes of bf16 per (db, mq), the
    // other half-wave holding the adjacent 8.  One permlane32 swap per pair (mq, mq + 1) and dword leaves lanes 0..31 with columns
    // 8 mq .. 8 mq + 7 and lanes 32..63 with 8 (mq + 1) .. + 7: 16-byte stores, half as many store instructions.
    const float inv = P.pscale / l;                               // 1/(1-p) of the dropout is applied here, once
    bf16* op = reinterpret_cast<bf16*>(P.out) + ((long long)b * P.T + q) * P.ld_o + h * DH + 8 * hh;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int mq = 0; mq < 4; mq += 2) {
            u32x2 a = __builtin_bit_cast(u32x2, (bf16x4){(bf16)(o[db][4 * mq] * inv), (bf16)(o[db][4 * mq + 1] * inv), (bf16)(o[db][4 * mq + 2] * inv), (bf16)(o[db][4 * mq + 3] * inv)});
            u32x2 c = __builtin_bit_cast(u32x2, (bf16x4){(bf16)(o[db][4 * mq + 4] * inv), (bf16)(o[db][4 * mq + 5] * inv), (bf16)(o[db][4 * mq + 6] * inv), (bf16)(o[db][4 * mq + 7] * inv)});
            auto x0 = __builtin_amdgcn_permlane32_swap(a[0], c[0], false, false);      // vdst = group mq, src = group mq + 1
            auto x1 = __builtin_amdgcn_permlane32_swap(a[1], c[1], false, false);
            // lanes 0..31: {own a | upper's a} = columns 8 mq + 0..7;  lanes 32..63: {lower's c | own c} = columns 8 (mq + 1) + 0..7
            const u32x4 v = {x0[0], x1[0], x0[1], x1[1]};
            if (q_ok) *reinterpret_cast<u32x4*>(op + 32 * db + 8 * mq) = v;
        }
    if (q_ok && hh == 0) P.lse[(long long)bh * P.T + q] = m + log2f(l);       // log2-domain log-sum-exp
}

// ================================================================== backward, 16-bit: 32x32x16 MFMA (round 6)
// The 16 x 16 backward kernels above read every staged operand tile twice per 16-row wave tile (row fragments for S / dP, transposed
// fragments for dV / dK or dQ: 0.75 KB of LDS reads per MFMA), stage through registers with a block barrier per tile, and fetch the
// softmax statistics of a key-stationary block from LDS per 16 x 16 fragment.  These two kernels are the forward's geometry
// (attn_fwd32_kernel) applied to the backward products:
//   * a wave owns 32 queries (dQ) / 32 keys (dK, dV) and multiplies on v_mfma_f32_32x32x16_bf16: half the LDS operand bytes per FLOP,
//     an MFMA blocks the vector issue for 8 of its 32 cycles instead of 8 of 16;
//   * dQ: S^T = K Q^T and dP^T = V dO^T with lane = query, so lse / delta are one register each and a dropout word covers four
//     consecutive registers (as in the forward); dK / dV: S = Q K^T and dP = dO V^T with lane = key, K and V stationary in registers as
//     the B operands; in both, the score registers 8 s .. 8 s + 7 ARE the B operand of the following product for 16-row step s;
//   * operand tiles arrive by LDS-DMA into a 3-deep ring (one barrier per 64-row tile, two tiles in flight, no staging registers);
//   * ONE LDS image per tile serves both read patterns: 128-byte rows, 16-byte chunk c of row r at c ^ swz3((r >> 1) & 7).  Row reads
//     (b128, 16 consecutive rows per quarter wave) need the keys of same-parity rows distinct: any bijection of (r >> 1) & 7.  Transposed
//     reads (ds_read_b64_tr_b16: a half-wave covers 8 consecutive rows x 64 bytes) need the keys' bit 2 to alternate over rows r, r + 2 of
//     an aligned group of 8: swz3 swaps bits 0 and 2.  (Round 3's 32 x 32 dQ kernel kept two K images and lost to the 16 x 16 one.)
//   * the transposed reads are inline assembly with their own lgkmcnt wait: in front of the builtin hipcc drains every LDS-DMA in flight.
// Dropout words, seeds and masks are those of the kernels above; results agree with them to bf16 rounding of P / dS (same products,
// another summation order).
A8_DEV int swz3(int t) { return ((t & 1) << 2) | (t & 2) | ((t >> 2) & 1); }
A8_DEV u32x2 lds_tr64(unsigned addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
A8_DEV void attn_glds4(const char* src, char* lds_dst_wave_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst_wave_uniform, 4, 0, 0);
}
A8_DEV uint32_t pack2(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, (bf16x2){(bf16)a, (bf16)b});
}
A8_DEV bf16x8 pack8(const f32x16& v, int o) {
    return (bf16x8){(bf16)v[o], (bf16)v[o + 1], (bf16)v[o + 2], (bf16)v[o + 3], (bf16)v[o + 4], (bf16)v[o + 5], (bf16)v[o + 6], (bf16)v[o + 7]};
}
// acc[db][i] = value of (row = lane & 31 of the wave's 32, column 32 db + (i & 3) + 8 (i >> 2) + 4 (lane >> 5)), scaled, as bf16 rows of 64:
// one permlane32 swap per register pair gives every lane 8 consecutive columns (16-byte stores), as in the forward's epilogue
A8_DEV void store_rows32(bf16* rowp /* + 8 * hh */, const f32x16 (&acc)[2], float sc, bool ok) {
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int mq = 0; mq < 4; mq += 2) {
            u32x2 a = __builtin_bit_cast(u32x2, (bf16x4){(bf16)(acc[db][4 * mq] * sc), (bf16)(acc[db][4 * mq + 1] * sc), (bf16)(acc[db][4 * mq + 2] * sc), (bf16)(acc[db][4 * mq + 3] * sc)});
            u32x2 c = __builtin_bit_cast(u32x2, (bf16x4){(bf16)(acc[db][4 * mq + 4] * sc), (bf16)(acc[db][4 * mq + 5] * sc), (bf16)(acc[db][4 * mq + 6] * sc), (bf16)(acc[db][4 * mq + 7] * sc)});
            auto x0 = __builtin_amdgcn_permlane32_swap(a[0], c[0], false, false);
            auto x1 = __builtin_amdgcn_permlane32_swap(a[1], c[1], false, false);
            const u32x4 v = {x0[0], x1[0], x0[1], x1[1]};
            if (ok) *reinterpret_cast<u32x4*>(rowp + 32 * db + 8 * mq) = v;
        }
}

// ---- dQ (query block stationary): streams K / V tiles of 64 keys
template <int NW, bool DROP>
__global__ __launch_bounds__(NW * 64, 3) void attn_bwd_q32_kernel(const AttnP P) {
    constexpr int BQ = NW * 32, NBUF = 3;
    __shared__ __attribute__((aligned(1024))) char lds[NBUF * F32_TILE + 1024];
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int blk, bh;
    block_coords((P.T + BQ - 1) / BQ, P.B * P.H, blk, bh);
    const int b = bh / P.H, h = bh % P.H;
    const bf16* qb = reinterpret_cast<const bf16*>(P.q) + (long long)b * P.T * P.ld_qkv + h * DH;
    const char* kb_ = reinterpret_cast<const char*>(reinterpret_cast<const bf16*>(P.k) + (long long)b * P.T * P.ld_qkv + h * DH);
    const char* vb_ = reinterpret_cast<const char*>(reinterpret_cast<const bf16*>(P.v) + (long long)b * P.T * P.ld_qkv + h * DH);
    const bf16* dob = reinterpret_cast<const bf16*>(P.dout) + (long long)b * P.T * P.ld_o + h * DH;
    const bf16* ob = reinterpret_cast<const bf16*>(P.o) + (long long)b * P.T * P.ld_o + h * DH;
    const uint8_t* km = P.keymask ? P.keymask + (long long)b * P.T : nullptr;
    const int ntiles = (P.T + BT - 1) / BT;
    const long long row_bytes = P.ld_qkv * 2, tile_bytes = (long long)BT * row_bytes;
    MaskScan ms;
    if (km) ms.load(km, P.T, lane);

    // ---- DMA pieces (1 KB = 8 rows each; 0..7 = K rows 8 p.., 8..15 = V rows 8 (p - 8)..): LDS slot lane % 8 of row 8 p + lane / 8 takes the
    // source chunk slot ^ swz3((row >> 1) & 7); rows beyond T are clamped to row T - 1 (their probabilities are zero)
    unsigned soff[4], soff_clamp[4]; int dst[4]; bool is_k[4];
    const int last_rows = P.T - (ntiles - 1) * BT;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pid = (wv + NW * j) & 15;
        const int row = 8 * (pid & 7) + (lane >> 3);
        const int chunk = (lane & 7) ^ swz3((row >> 1) & 7);
        is_k[j] = pid < 8;
        soff[j] = (unsigned)(row * (int)row_bytes + chunk * 16);
        soff_clamp[j] = (unsigned)((row - min(row, last_rows - 1)) * (int)row_bytes);
        dst[j] = pid * 1024;
    }
    auto issue = [&](int t, int slot, int nlive) __attribute__((always_inline)) {
        const bool live = t < nlive;
        const int tt = live ? t : ntiles - 1;
        const char* kbase = kb_ + (long long)tt * tile_bytes;
        const char* vbase = vb_ + (long long)tt * tile_bytes;
        char* base = live ? lds + slot * F32_TILE : lds + NBUF * F32_TILE;
        const bool lastt = tt == ntiles - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            attn_glds16((is_k[j] ? kbase : vbase) + (soff[j] - (lastt ? soff_clamp[j] : 0u)), base + (live ? dst[j] : 0));
    };

    // ---- this lane's query: Q and dO fragments (B operands: [query][16 ks + 8 half + 0..7]), O for delta, lse
    const int q = blk * BQ + wv * 32 + r;
    const bool q_ok = q < P.T;
    const bool wave_live = blk * BQ + wv * 32 < P.T;
    bf16x8 qf[4], dof[4], of[4];
    const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        qf[ks] = q_ok ? *reinterpret_cast<const bf16x8*>(qb + (long long)q * P.ld_qkv + 16 * ks + 8 * hh) : z8;
        dof[ks] = q_ok ? *reinterpret_cast<const bf16x8*>(dob + (long long)q * P.ld_o + 16 * ks + 8 * hh) : z8;
        of[ks] = q_ok ? *reinterpret_cast<const bf16x8*>(ob + (long long)q * P.ld_o + 16 * ks + 8 * hh) : z8;
    }
    const float lse_q = q_ok ? P.lse[(long long)bh * P.T + q] : 0.f;
    issue(0, 0, ntiles);
    issue(1, 1, ntiles);
    __builtin_amdgcn_s_waitcnt(0x0F78);          // vmcnt(8): everything older than the eight DMAs (see attn_fwd32_kernel)
    const int nv = km ? ms.finish(lane) : 1;
    const bool holes = km != nullptr && nv <= 0;
    const float maskv = mask_fill_value(nv);
    const int Tk = (km != nullptr && nv > 0) ? nv : P.T, ntl = (Tk + BT - 1) / BT;
    // delta = rowsum(dO * O) of this lane's query (its two half-waves hold complementary halves of the head dimension); published for
    // the dK / dV kernel in units of the dropout scale -- stored at the END (a store in flight would sit in the DMA ring's counted waits)
    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) dl = __builtin_fmaf((float)of[ks][e], (float)dof[ks][e], dl);
    {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(dl), __float_as_uint(dl), false, false);
        dl = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    const float nlse = -lse_q, ndelta = -dl;
    f32x16 dq[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) dq[db][i] = 0.f;
    uint32_t hbase = (pair_base(bh + P.bh0, P.T, P.Th, q) + (uint32_t)hh + P.seed) * HC1;
    // row reads: row 32 kb + r, chunk 2 ks + half
    int kaddr[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kaddr[ks] = r * 128 + (((2 * ks + hh) ^ swz3((r >> 1) & 7)) << 4);
    // transposed reads of the K image (A = K^T: lane's d = 32 db + 16 mh + i, k-slots = keys 16 s + 4 half + (0..3 | 8..11)):
    // lane (i = lane & 15, mh = (lane >> 4) & 1) reads 8 bytes of row 4 half + i / 4 (+ 8), columns 32 db + 16 mh + 4 (i & 3) ..
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    unsigned tr1[2], tr2[2];
    {
        const int i = lane & 15, mh = (lane >> 4) & 1, rowA = 4 * hh + (i >> 2), p = i & 3;
        const int f1 = swz3(rowA >> 1), f2 = swz3((rowA >> 1) + 4);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const int c = 4 * db + 2 * mh + (p >> 1);
            tr1[db] = lds0 + (unsigned)(rowA * 128 + ((c ^ f1) << 4) + (p & 1) * 8);
            tr2[db] = lds0 + (unsigned)((rowA + 8) * 128 + ((c ^ f2) << 4) + (p & 1) * 8);
        }
    }

    int slot = 0;
    for (int t = 0; t < ntl; ++t) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int nslot = slot + 2; nslot = nslot >= NBUF ? nslot - NBUF : nslot;
        const int toff = slot * F32_TILE;
        const char* kt = lds + toff;
        const char* vt = kt + 8192;
        slot = slot + 1 == NBUF ? 0 : slot + 1;
        if (wave_live) {
            const bool edge = holes || (t + 1) * BT > Tk;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                f32x16 s, dp;
#pragma unroll
                for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) s = mfma32(*reinterpret_cast<const bf16x8*>(kt + kb * 4096 + kaddr[ks]), qf[ks], s);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) dp = mfma32(*reinterpret_cast<const bf16x8*>(vt + kb * 4096 + kaddr[ks]), dof[ks], dp);
                // the K^T fragments of this key block for dQ: requested now, consumed behind the softmax arithmetic
                u32x2 ka[2][2][2];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        ka[s2][db][0] = lds_tr64(tr1[db] + (unsigned)(toff + kb * 4096 + s2 * 2048));
                        ka[s2][db][1] = lds_tr64(tr2[db] + (unsigned)(toff + kb * 4096 + s2 * 2048));
                    }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float x = __builtin_fmaf(s[i], P.scale, nlse);
                    bool filled = false;
                    if (edge) {
                        const int key = t * BT + 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * hh;
                        if (holes) {
                            filled = key >= P.T || !km[min(key, P.T - 1)];
                            if (filled) x = maskv + nlse;
                        } else if (key >= Tk) x = -INFINITY;
                    }
                    const float p = fast_exp2(x);
                    float dpv = dp[i];
                    if (DROP) {
                        // one hash word per key quad = 4 consecutive registers (the forward's words)
                        const uint32_t h0 = hfin(hbase + (uint32_t)(8 * kb + 2 * (i >> 2)) * HC1);
                        dpv = (((h0 >> (8 * (i & 3))) & 0xFFu) >= P.thr) ? dpv : 0.f;
                    }
                    s[i] = (edge && filled) ? 0.f : p * __builtin_fmaf(dpv, P.pscale, ndelta);      // dS^T[key][q] / softmax scale
                }
                const bf16x8 dsf[2] = {pack8(s, 0), pack8(s, 8)};
                __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): the transposed reads above
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const u32x4 av = {ka[s2][db][0][0], ka[s2][db][0][1], ka[s2][db][1][0], ka[s2][db][1][1]};
                        dq[db] = mfma32(__builtin_bit_cast(bf16x8, av), dsf[s2], dq[db]);
                    }
            }
            hbase += (uint32_t)(BT / 4) * HC1;
        }
        issue(t + 2, nslot, ntl);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (q_ok && hh == 0) P.delta[(long long)bh * P.T + q] = dl * P.ipscale;
    store_rows32(reinterpret_cast<bf16*>(P.dq) + ((long long)b * P.T + q) * P.ld_d + h * DH + 8 * hh, dq, P.dscale, q_ok);
}

// ---- dK, dV (key block stationary): streams Q / dO tiles of 64 queries + their lse / delta
constexpr int KV32_TILE = 17408;                // Q image 8 KB | dO image 8 KB | lse[64] delta[64] f32 | pad
template <int NW, bool DROP>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_kv32_kernel(const AttnP P) {
    constexpr int BK = NW * 32, NBUF = 3;
    __shared__ __attribute__((aligned(1024))) char lds[NBUF * KV32_TILE + 1024];
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int blk, bh;
    block_coords((P.T + BK - 1) / BK, P.B * P.H, blk, bh);
    const int b = bh / P.H, h = bh % P.H;
    const char* qb_ = reinterpret_cast<const char*>(reinterpret_cast<const bf16*>(P.q) + (long long)b * P.T * P.ld_qkv + h * DH);
    const bf16* kb = reinterpret_cast<const bf16*>(P.k) + (long long)b * P.T * P.ld_qkv + h * DH;
    const bf16* vb = reinterpret_cast<const bf16*>(P.v) + (long long)b * P.T * P.ld_qkv + h * DH;
    const char* dob_ = reinterpret_cast<const char*>(reinterpret_cast<const bf16*>(P.dout) + (long long)b * P.T * P.ld_o + h * DH);
    const float* lseb = P.lse + (long long)bh * P.T;
    const float* delb = P.delta + (long long)bh * P.T;
    const int ntiles = (P.T + BT - 1) / BT;
    const int key = blk * BK + wv * 32 + r;
    const bool k_ok = key < P.T;
    const bool wave_live = blk * BK + wv * 32 < P.T;
    float maskv = MASKED_LOG2;
    bool key_masked = false;
    if (P.keymask) {
        // prefix key mask: blocks beyond the attended prefix store zeros (see attn_bwd_kv_kernel)
        const int nv = mask_prefix_len(P.keymask + (long long)b * P.T, P.T, lane);
        maskv = mask_fill_value(nv);
        key_masked = k_ok && !P.keymask[(long long)b * P.T + key];
        if (nv > 0 && blk * BK >= nv) {
            f32x16 zz[2];
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int i = 0; i < 16; ++i) zz[db][i] = 0.f;
            store_rows32(reinterpret_cast<bf16*>(P.dk) + ((long long)b * P.T + key) * P.ld_d + h * DH + 8 * hh, zz, 1.f, k_ok);
            store_rows32(reinterpret_cast<bf16*>(P.dv) + ((long long)b * P.T + key) * P.ld_d + h * DH + 8 * hh, zz, 1.f, k_ok);
            return;
        }
    }
    const long long qrow_bytes = P.ld_qkv * 2, orow_bytes = P.ld_o * 2;

    // ---- DMA pieces: 0..7 = Q rows 8 p.., 8..15 = dO rows 8 (p - 8)..; plus one 4-byte-per-lane piece per wave: lse (even waves) /
    // delta (odd waves) of the tile's 64 queries (two waves stage each: same bytes, same place -- every wave issues 5 pieces per tile).
    // Wave wv stages pieces wv, wv + 4 (Q) and wv + 8, wv + 12 (dO): their rows' swizzle keys ((row >> 1) & 7 = 4 (piece & 1) + lane / 16)
    // are the same, so a lane's source offset is ONE register per operand; the last tile's row clamp is formed when it is needed.
    static_assert(NW == 4, "piece map");
    const int last_rows = P.T - (ntiles - 1) * BT;
    const int lrow = lane >> 3;
    const int chunk = (lane & 7) ^ swz3((4 * (wv & 1) + (lane >> 4)) & 7);
    const unsigned qlane = (unsigned)(lrow * (int)qrow_bytes + chunk * 16), olane = (unsigned)(lrow * (int)orow_bytes + chunk * 16);
    const float* statb = (wv & 1) ? delb : lseb;
    const int stat_dst = 16384 + (wv & 1) * 256;
    auto issue = [&](int t, int slot, int nlive) __attribute__((always_inline)) {
        const bool live = t < nlive;
        const int tt = live ? t : ntiles - 1;
        const char* qbase = qb_ + (long long)tt * BT * qrow_bytes;
        const char* obase = dob_ + (long long)tt * BT * orow_bytes;
        char* base = live ? lds + slot * KV32_TILE : lds + NBUF * KV32_TILE;
        const bool lastt = tt == ntiles - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pid = wv + 4 * j, prow = 8 * (pid & 7);
            const long long rb = j < 2 ? qrow_bytes : orow_bytes;
            long long off = (long long)prow * rb + (j < 2 ? qlane : olane);       // (signed: the clamp may exceed the lane's own part)
            if (lastt) off -= (long long)max(0, prow + lrow - (last_rows - 1)) * rb;
            attn_glds16((j < 2 ? qbase : obase) + off, base + (live ? pid * 1024 : 0));
        }
        attn_glds4(reinterpret_cast<const char*>(statb + min(tt * BT + lane, P.T - 1)), base + (live ? stat_dst : 0));
    };

    // ---- this lane's key: K and V fragments (B operands: [key][16 ks + 8 half + 0..7])
    bf16x8 kf[4], vf[4];
    const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        kf[ks] = k_ok ? *reinterpret_cast<const bf16x8*>(kb + (long long)key * P.ld_qkv + 16 * ks + 8 * hh) : z8;
        vf[ks] = k_ok ? *reinterpret_cast<const bf16x8*>(vb + (long long)key * P.ld_qkv + 16 * ks + 8 * hh) : z8;
    }
    issue(0, 0, ntiles);
    issue(1, 1, ntiles);
    __builtin_amdgcn_s_waitcnt(0x0F7A);          // vmcnt(10): the K / V fragment loads are older than the ten DMAs
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dk[db][i] = 0.f; dv[db][i] = 0.f; }
    // Dropout word of (query, key quad): the four lanes of a quad hold the four keys of one key quad and need the same 16 words per 32
    // queries; lane (key & 3) = r4 hashes the words of queries 4 m + r4 (+ 4 half) only, the quad exchanges them by DPP (as the 16 x 16 kernel)
    const uint32_t ThC = (uint32_t)P.Th * HC1;
    uint32_t hk = (pair_base(bh + P.bh0, P.T, P.Th, 4 * hh + (lane & 3)) + (uint32_t)(key >> 2) + P.seed) * HC1;
    const uint32_t fmask = 0xFFu << (8 * (lane & 3)), fthr = P.thr << (8 * (lane & 3));
    int raddr[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) raddr[ks] = r * 128 + (((2 * ks + hh) ^ swz3((r >> 1) & 7)) << 4);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    unsigned tr1[2], tr2[2];
    {
        const int i = lane & 15, mh = (lane >> 4) & 1, rowA = 4 * hh + (i >> 2), p = i & 3;
        const int f1 = swz3(rowA >> 1), f2 = swz3((rowA >> 1) + 4);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const int c = 4 * db + 2 * mh + (p >> 1);
            tr1[db] = lds0 + (unsigned)(rowA * 128 + ((c ^ f1) << 4) + (p & 1) * 8);
            tr2[db] = lds0 + (unsigned)((rowA + 8) * 128 + ((c ^ f2) << 4) + (p & 1) * 8);
        }
    }

    int slot = 0;
    for (int t = 0; t < ntiles; ++t) {
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int nslot = slot + 2; nslot = nslot >= NBUF ? nslot - NBUF : nslot;
        const int toff = slot * KV32_TILE;
        const char* qt = lds + toff;
        const char* dot = qt + 8192;
        const float* stat = reinterpret_cast<const float*>(qt + 16384);
        slot = slot + 1 == NBUF ? 0 : slot + 1;
        if (wave_live) {
            const bool edge = (t + 1) * BT > P.T;                 // wave-uniform: the last tile's rows beyond T are copies of row T - 1
#pragma unroll
            for (int qb2 = 0; qb2 < 2; ++qb2) {
                // Four phases per 32 queries, ordered for register pressure (K, V fragments and both accumulators are 96 registers):
                // S -> P -> dV += dO^T P, then dP -> dS -> dK += Q^T dS; each transposed operand is requested one phase ahead of its MFMAs.
                const unsigned o = (unsigned)(toff + qb2 * 4096);
                f32x16 s;
#pragma unroll
                for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) s = mfma32(*reinterpret_cast<const bf16x8*>(qt + qb2 * 4096 + raddr[ks]), kf[ks], s);
                u32x2 oa[2][2][2];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        oa[s2][db][0] = lds_tr64(tr1[db] + o + (unsigned)(s2 * 2048) + 8192u);
                        oa[s2][db][1] = lds_tr64(tr2[db] + o + (unsigned)(s2 * 2048) + 8192u);
                    }
                uint32_t pw[8];                                   // P as bf16 pairs: registers 2 w, 2 w + 1
                uint32_t keep = 0xFFFFu;                          // bit i: score register i survives the dropout
#pragma unroll
                for (int m = 0; m < 4; ++m) {                     // queries 32 qb2 + 8 m + 4 half + 0..3 = registers 4 m .. 4 m + 3
                    const f32x4 lse4 = *reinterpret_cast<const f32x4*>(stat + 32 * qb2 + 8 * m + 4 * hh);
                    uint32_t w = 0;
                    if (DROP) w = hfin(hk + (uint32_t)(32 * qb2 + 8 * m) * ThC);
                    const uint32_t wr[4] = {quad_bcast<0>(w), quad_bcast<1>(w), quad_bcast<2>(w), quad_bcast<3>(w)};
                    float pq[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = 4 * m + e;
                        float x = __builtin_fmaf(s[i], P.scale, -lse4[e]);
                        if (key_masked) x = maskv - lse4[e];
                        if (edge && t * BT + 32 * qb2 + 8 * m + 4 * hh + e >= P.T) x = -INFINITY;
                        const float p = fast_exp2(x);
                        s[i] = p;
                        bool kp = true;
                        if (DROP) { kp = (wr[e] & fmask) >= fthr; if (!kp) keep &= ~(1u << i); }
                        pq[e] = kp ? p : 0.f;                                             // P (dropped, unscaled) for dV
                    }
                    pw[2 * m] = pack2(pq[0], pq[1]);
                    pw[2 * m + 1] = pack2(pq[2], pq[3]);
                }
                const bf16x8 pf[2] = {__builtin_bit_cast(bf16x8, (u32x4){pw[0], pw[1], pw[2], pw[3]}), __builtin_bit_cast(bf16x8, (u32x4){pw[4], pw[5], pw[6], pw[7]})};
                __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): the dO^T fragments
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const u32x4 ov = {oa[s2][db][0][0], oa[s2][db][0][1], oa[s2][db][1][0], oa[s2][db][1][1]};
                        dv[db] = mfma32(__builtin_bit_cast(bf16x8, ov), pf[s2], dv[db]);
                    }
                f32x16 dp;
#pragma unroll
                for (int i = 0; i < 16; ++i) dp[i] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) dp = mfma32(*reinterpret_cast<const bf16x8*>(dot + qb2 * 4096 + raddr[ks]), vf[ks], dp);
                uint32_t dw[8];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const f32x4 del4 = *reinterpret_cast<const f32x4*>(stat + 64 + 32 * qb2 + 8 * m + 4 * hh);      // delta / pscale (see the dQ kernel)
                    float dq4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = 4 * m + e;
                        const float dpv = (!DROP || ((keep >> i) & 1u)) ? dp[i] : 0.f;
                        float ds = s[i] * (dpv - del4[e]);                                // dS / (softmax scale * pscale) for dK
                        if (key_masked) ds = 0.f;                                         // a filled score does not depend on q or k
                        dq4[e] = ds;
                    }
                    dw[2 * m] = pack2(dq4[0], dq4[1]);
                    dw[2 * m + 1] = pack2(dq4[2], dq4[3]);
                }
                const bf16x8 dsf[2] = {__builtin_bit_cast(bf16x8, (u32x4){dw[0], dw[1], dw[2], dw[3]}), __builtin_bit_cast(bf16x8, (u32x4){dw[4], dw[5], dw[6], dw[7]})};
                // (requested only now: sixteen more registers across the dS arithmetic spill the K / V fragments at 3 waves per SIMD)
                __builtin_amdgcn_sched_barrier(0);
                u32x2 qa[2][2][2];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        qa[s2][db][0] = lds_tr64(tr1[db] + o + (unsigned)(s2 * 2048));
                        qa[s2][db][1] = lds_tr64(tr2[db] + o + (unsigned)(s2 * 2048));
                    }
                __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): the Q^T fragments
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int db = 0; db < 2; ++db) {
                        const u32x4 qv = {qa[s2][db][0][0], qa[s2][db][0][1], qa[s2][db][1][0], qa[s2][db][1][1]};
                        dk[db] = mfma32(__builtin_bit_cast(bf16x8, qv), dsf[s2], dk[db]);
                    }
            }
            hk += (uint32_t)BT * ThC;
        }
        issue(t + 2, nslot, ntiles);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    store_rows32(reinterpret_cast<bf16*>(P.dk) + ((long long)b * P.T + key) * P.ld_d + h * DH + 8 * hh, dk, P.dscale * P.pscale, k_ok);
    store_rows32(reinterpret_cast<bf16*>(P.dv) + ((long long)b * P.T + key) * P.ld_d + h * DH + 8 * hh, dv, P.pscale, k_ok);
}

void set_softmax_params(AttnP& P, float scale, float p, uint32_t seed) {
    P.scale = scale * 1.4426950408889634f;
    P.dscale = scale;
    P.lazy = 8.0f;
    P.Th = (P.T + 3) / 4;
    uint32_t thr = p > 0.f ? (uint32_t)(p * 256.0f + 0.5f) : 0u;      // the drop probability is realised as thr / 256
    if (p > 0.f && thr == 0) thr = 1;
    if (thr > 255u) thr = 255u;
    P.thr = thr;
    P.pscale = thr ? 256.0f / (256.0f - (float)thr) : 1.0f;            // unbiased for the realised keep probability
    P.ipscale = (256.0f - (float)thr) / 256.0f;
    P.seed = seed;
}

int check_common(const char* what, int dtype, int B, int T_, int H, long long ld_qkv, long long ld_o, float p) {
    A8_REQUIRE(dtype == A8_F32 || dtype == A8_BF16, "%s: bad dtype", what);
    A8_REQUIRE(B >= 1 && T_ >= 1 && H >= 1, "%s: bad sizes B=%d T=%d H=%d", what, B, T_, H);
    const int epc = dtype == A8_BF16 ? 8 : 4;
    A8_REQUIRE(ld_qkv % epc == 0 && ld_o % epc == 0 && ld_qkv >= (long long)H * DH && ld_o >= (long long)H * DH, "%s: bad leading dimensions", what);
    A8_REQUIRE(p >= 0.f && p < 1.f, "%s: bad dropout p", what);
    A8_REQUIRE((long long)B * H * T_ * ((T_ + 3) / 4) < (1ll << 32), "%s: B*H*T*T/4 exceeds the dropout counter range", what);
    return A8_OK;
}

// launch geometry (a8_set_tuning keys 2 = forward, 3 = dK/dV, 4 = dQ): value = waves_per_block * 10 + row groups per wave
// (any hundreds digit: the same geometry through this generic macro, i.e. with run-time dropout / mask flags -- 141 is the A/B partner
// of the flag-specialised default 41 of the backward kernels)
#define A8_ATTN_LAUNCH(KERN, key, what)                                                                                       \
    {                                                                                                                          \
        const int tun = g_a8_tuning[key] >= 10 ? g_a8_tuning[key] : 41;            /* (< 10 selects the 32x32 16-bit forward kernel; f32 data: 4 x 1) */ \
        const int geo = tun % 100, nw = geo / 10, grp = geo % 10, rows = nw * grp * 16;                                       \
        const dim3 grid(cdiv(T_, rows) * B * H), blk(nw * 64);                                                                 \
        if (dtype == A8_BF16) {                                                                                                \
            if (geo == 41) hipLaunchKernelGGL((KERN<bf16, 4, 1>), grid, blk, 0, s, P);                                        \
            else if (geo == 22) hipLaunchKernelGGL((KERN<bf16, 2, 2>), grid, blk, 0, s, P);                                   \
            else if (geo == 42) hipLaunchKernelGGL((KERN<bf16, 4, 2>), grid, blk, 0, s, P);                                   \
            else return a8_set_error(A8_ERR_ARG, "%s: bad tuning %d", what, geo);                                             \
        } else {                                                                                                               \
            const dim3 grid1(cdiv(T_, 64) * B * H);                                                                            \
            hipLaunchKernelGGL((KERN<float, 4, 1>), grid1, dim3(256), 0, s, P);                                               \
        }                                                                                                                      \
    }

// the default 4 x 1 geometry of the 16-bit backward kernels, dropout / key mask resolved at compile time
#define A8_ATTN_LAUNCH_FLAGS(KERN, NW_, G_, W)                                                                                           \
    {                                                                                                                          \
        const dim3 grid(cdiv(T_, NW_ * G_ * 16) * B * H), blk(NW_ * 64);                                                      \
        switch (flags) {                                                                                                       \
            case 0: hipLaunchKernelGGL((KERN<bf16, NW_, G_, W, 0, 0>), grid, blk, 0, s, P); break;                               \
            case 1: hipLaunchKernelGGL((KERN<bf16, NW_, G_, W, 1, 0>), grid, blk, 0, s, P); break;                               \
            case 2: hipLaunchKernelGGL((KERN<bf16, NW_, G_, W, 0, 1>), grid, blk, 0, s, P); break;                               \
            default: hipLaunchKernelGGL((KERN<bf16, NW_, G_, W, 1, 1>), grid, blk, 0, s, P); break;                              \
        }                                                                                                                      \
    }

}  // namespace

int a8i_attn_fwd(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* keymask,
                 int B, int T_, int H, int64_t ld_qkv, int64_t ld_o, float scale, float p, uint32_t seed, int bh0, void* stream) {
    if (int e = check_common("a8_attn_fwd", dtype, B, T_, H, ld_qkv, ld_o, p)) return e;
    AttnP P = {};
    P.q = (const char*)q; P.k = (const char*)k; P.v = (const char*)v; P.out = (char*)o; P.lse = lse; P.keymask = keymask;
    P.B = B; P.T = T_; P.H = H; P.ld_qkv = ld_qkv; P.ld_o = ld_o;
    set_softmax_params(P, scale, p, seed);
    P.bh0 = bh0;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == A8_BF16 && g_a8_tuning[2] < 10) {
        // 32 queries per wave, 4 or 5 waves per block: whichever wastes fewer wave slots on the tail of T (781 = 24.4 x 32 -> 5 x 5 waves)
        const int units = cdiv(T_, 32);
        const int nw = g_a8_tuning[2] == 5 ? 5 : 4;       // (4 measured faster than 5 even at T = 781 = 24.4 x 32, where 5 x 5 waves waste none)
        (void)units;
        const dim3 grid(cdiv(T_, 32 * nw) * B * H);
        if (nw == 5) hipLaunchKernelGGL(attn_fwd32_kernel<5>, grid, dim3(320), 0, s, P);
        else hipLaunchKernelGGL(attn_fwd32_kernel<4>, grid, dim3(256), 0, s, P);
        return a8_check_launch("a8_attn_fwd");
    }
    A8_ATTN_LAUNCH(attn_fwd_kernel, 2, "a8_attn_fwd")
    return a8_check_launch("a8_attn_fwd");
}

extern "C" int a8_attn_fwd(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* keymask,
                           int B, int T_, int H, int64_t ld_qkv, int64_t ld_o, float scale, float p, uint32_t seed, void* stream) {
    return a8i_attn_fwd(dtype, q, k, v, o, lse, keymask, B, T_, H, ld_qkv, ld_o, scale, p, seed, 0, stream);
}

int a8i_attn_bwd(int dtype, const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                 const uint8_t* keymask, void* dq, void* dk, void* dv, float* delta, float* dbias, int B, int T_, int H, int64_t ld_qkv,
                 int64_t ld_o, int64_t ld_dqkv, float scale, float p, uint32_t seed, int bh0, void* stream) {
    if (int e = check_common("a8_attn_bwd", dtype, B, T_, H, ld_qkv, ld_o, p)) return e;
    AttnP P = {};
    P.q = (const char*)q; P.k = (const char*)k; P.v = (const char*)v; P.o = (const char*)o; P.dout = (const char*)dout;
    P.lse = (float*)lse; P.delta = delta; P.dbias = dbias; P.keymask = keymask;
    P.dq = (char*)dq; P.dk = (char*)dk; P.dv = (char*)dv;
    P.B = B; P.T = T_; P.H = H; P.ld_qkv = ld_qkv; P.ld_o = ld_o; P.ld_d = ld_dqkv;
    set_softmax_params(P, scale, p, seed);
    P.bh0 = bh0;
    hipStream_t s = (hipStream_t)stream;
    // dQ first: it computes delta = rowsum(dO * O) for its own query rows and publishes it for the dK/dV kernel
    const int flags = (P.thr != 0 ? 1 : 0) + (keymask ? 2 : 0);
    // a8_set_tuning keys 4 (dQ) / 3 (dK, dV) = 32: the 32 x 32-MFMA kernels of round 6 (16-bit data, no fused bias gradients)
    const bool can32 = dtype == A8_BF16 && dbias == nullptr && ld_qkv % 8 == 0 && ld_o % 8 == 0 && ld_dqkv % 8 == 0;
    if (can32 && g_a8_tuning[4] == 32) {
        const dim3 grid(cdiv(T_, 128) * B * H);
        if (P.thr) hipLaunchKernelGGL((attn_bwd_q32_kernel<4, true>), grid, dim3(256), 0, s, P);
        else hipLaunchKernelGGL((attn_bwd_q32_kernel<4, false>), grid, dim3(256), 0, s, P);
    }
    else if (dtype == A8_BF16 && (g_a8_tuning[4] == 41 || g_a8_tuning[4] == 32)) A8_ATTN_LAUNCH_FLAGS(attn_bwd_q_kernel, 4, 1, 1)
    else A8_ATTN_LAUNCH(attn_bwd_q_kernel, 4, "a8_attn_bwd")
    if (can32 && g_a8_tuning[3] == 32) {
        const dim3 grid(cdiv(T_, 128) * B * H);
        if (P.thr) hipLaunchKernelGGL((attn_bwd_kv32_kernel<4, true>), grid, dim3(256), 0, s, P);
        else hipLaunchKernelGGL((attn_bwd_kv32_kernel<4, false>), grid, dim3(256), 0, s, P);
    }
    else if (dtype == A8_BF16 && (g_a8_tuning[3] == 41 || g_a8_tuning[3] == 32)) A8_ATTN_LAUNCH_FLAGS(attn_bwd_kv_kernel, 4, 1, 3)
    else A8_ATTN_LAUNCH(attn_bwd_kv_kernel, 3, "a8_attn_bwd")
    return a8_check_launch("a8_attn_bwd");
}

extern "C" int a8_attn_bwd(int dtype, const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                           const uint8_t* keymask, void* dq, void* dk, void* dv, float* delta, float* dbias, int B, int T_, int H, int64_t ld_qkv,
                           int64_t ld_o, int64_t ld_dqkv, float scale, float p, uint32_t seed, void* stream) {
    return a8i_attn_bwd(dtype, q, k, v, o, dout, lse, keymask, dq, dk, dv, delta, dbias, B, T_, H, ld_qkv, ld_o, ld_dqkv, scale, p, seed, 0, stream);
}
