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
//    word per PAIR of adjacent keys, h = fin((pair_index + seed) * C1), a 16-bit field per key.  The pair index
//    is affine in the tile / fragment coordinates, so the product is carried as a running sum (one add per word).
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
template <typename T> struct ATile;
template <> struct ATile<bf16> {
    static constexpr int BYTES = BT * 128;
    static constexpr int CPR = 8;                       // 16-byte chunks per row
    static A8_DEV int off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
    static A8_DEV bf16x8 row_frag(const char* t, int rb, int ks, int lane) {
        const int row = rb + (lane & 15);
        return *reinterpret_cast<const bf16x8*>(t + off(row, ks * 4 + (lane >> 4)));
    }
    // acc += X[rows kb..kb+31 in accumulator order][cols c0..c0+15]^T-contracted with the accumulator pair (p0,p1):
    //   acc[reg r][lane] = out[c0 + 4g + r][lane&15] += sum_rows X[row][c0 + 4g + r] * P[row][lane&15]
    static A8_DEV f32x4 mma_xt_p(const char* t, int kb, int c0, f32x4 p0, f32x4 p1, f32x4 acc, int lane) {
        const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
        const int r0 = kb + 4 * g + q, r1 = r0 + 16;
        const int ch = (c0 >> 3) + (p >> 1), sub = (p & 1) << 3;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, t + off(r0, ch) + sub));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, t + off(r1, ch) + sub));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        s16x8 xv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        bf16x8 pv = {(bf16)p0[0], (bf16)p0[1], (bf16)p0[2], (bf16)p0[3], (bf16)p1[0], (bf16)p1[1], (bf16)p1[2], (bf16)p1[3]};
        return Mma<bf16>::mma(__builtin_bit_cast(bf16x8, xv), pv, acc);
    }
};
template <> struct ATile<float> {
    static constexpr int LDF = 68;                      // floats per row (272 B, 16-byte aligned)
    static constexpr int BYTES = BT * LDF * 4;
    static constexpr int CPR = 16;
    static A8_DEV int off(int row, int chunk) { return (row * LDF + chunk * 4) * 4; }
    static A8_DEV float row_frag(const char* t, int rb, int ks, int lane) {
        return *reinterpret_cast<const float*>(t + ((rb + (lane & 15)) * LDF + ks * 4 + (lane >> 4)) * 4);
    }
    static A8_DEV f32x4 mma_xt_p(const char* t, int kb, int c0, f32x4 p0, f32x4 p1, f32x4 acc, int lane) {
        const int g = lane >> 4, i = lane & 15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = *reinterpret_cast<const float*>(t + ((kb + 4 * g + r) * LDF + c0 + i) * 4);
            acc = Mma<float>::mma(a, p0[r], acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = *reinterpret_cast<const float*>(t + ((kb + 16 + 4 * g + r) * LDF + c0 + i) * 4);
            acc = Mma<float>::mma(a, p1[r], acc);
        }
        return acc;
    }
};

template <typename T> struct AStage {               // global -> registers -> LDS for one 64x64 tile, 256 threads
    static constexpr int CPR = ATile<T>::CPR;
    static constexpr int NCH = BT * CPR / 256;       // chunks per thread (bf16: 2, f32: 4)
    static constexpr int RSTEP = 256 / CPR;
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

// S^T tile block: acc[i] (i = 0..3) = rows 16i.. of (Tile . RowFrags^T): acc[i][r] = sum_d Tile[16i+4g+r][d] * row[lane&15][d]
template <typename T> A8_DEV void tile_times_rows(const char* tile, const RowFrags<T>& rf, f32x4 (&acc)[4], int lane) {
    constexpr int NKS = DH / Mma<T>::KI;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = Mma<T>::mma(ATile<T>::row_frag(tile, 16 * i, ks, lane), rf.f[ks], acc[i]);
}
// the mirrored product: acc[i][r] = sum_d row... with the tile as the B-side: out[q = 16i+4g+r][key = lane&15]
template <typename T> A8_DEV void rows_times_tile(const char* tile, const RowFrags<T>& rf, f32x4 (&acc)[4], int lane) {
    constexpr int NKS = DH / Mma<T>::KI;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = Mma<T>::mma(ATile<T>::row_frag(tile, 16 * i, ks, lane), rf.f[ks], acc[i]);
}

struct AttnP {
    const char *q, *k, *v, *o, *dout;
    char *out, *dq, *dk, *dv;
    float *lse, *delta;
    const uint8_t* keymask;
    int B, T, H;
    long long ld_qkv, ld_o, ld_d;
    float scale, pscale, dscale;   // scale = softmax scale * log2(e) (scores and lse live in the log2 domain); dscale = softmax scale
    float lazy;                    // raise the running max only past this margin (log2 units)
    uint32_t thr, seed;            // thr: 16-bit keep threshold (0 = dropout off)
    int Th;                        // key pairs per query row = ceil(T / 2)
};

constexpr uint32_t HC1 = 0x9E3779B1u, HC2 = 0x2C1B3C6Du;
constexpr float MASKED_LOG2 = -1.4426950408889634e9f;      // masked_fill(-1e9) in the log2 domain
// dropout word of key pair `idx`: fin((idx + seed) * HC1); callers carry the product as a running sum
A8_DEV uint32_t hfin(uint32_t h) { h ^= h >> 15; return h * HC2; }
A8_DEV uint32_t pair_base(int bh, int T, int Th, int q) { return ((uint32_t)bh * (uint32_t)T + (uint32_t)q) * (uint32_t)Th; }
A8_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

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

template <typename T> A8_DEV void store4(T* p, float a, float b, float c, float d);
template <> A8_DEV void store4<bf16>(bf16* p, float a, float b, float c, float d) {
    *reinterpret_cast<bf16x4*>(p) = (bf16x4){(bf16)a, (bf16)b, (bf16)c, (bf16)d};
}
template <> A8_DEV void store4<float>(float* p, float a, float b, float c, float d) { *reinterpret_cast<f32x4*>(p) = (f32x4){a, b, c, d}; }

// ================================================================== forward
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnP P) {
    __shared__ __attribute__((aligned(16))) char lds[2 * ATile<T>::BYTES];
    char* kt = lds;
    char* vt = lds + ATile<T>::BYTES;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
    int blk, bh;
    block_coords((P.T + BT - 1) / BT, P.B * P.H, blk, bh);
    const int b = bh / P.H, h = bh % P.H;
    const int q0 = blk * BT + wv * 16, qi = q0 + li;
    const T* qb = reinterpret_cast<const T*>(P.q) + (long long)b * P.T * P.ld_qkv + h * DH;
    const T* kb = reinterpret_cast<const T*>(P.k) + (long long)b * P.T * P.ld_qkv + h * DH;
    const T* vb = reinterpret_cast<const T*>(P.v) + (long long)b * P.T * P.ld_qkv + h * DH;
    RowFrags<T> qf;
    qf.load(qb + (long long)qi * P.ld_qkv, qi < P.T, lane);
    const uint8_t* km = P.keymask ? P.keymask + (long long)b * P.T : nullptr;

    f32x4 oacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) oacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;                     // m: the max the probabilities are referred to (log2 domain)
    const int ntiles = (P.T + BT - 1) / BT;
    uint32_t hb = (pair_base(bh, P.T, P.Th, qi) + (uint32_t)(2 * g) + P.seed) * HC1;    // dropout word base of this tile
    AStage<T> sk, sv;
    sk.load(kb, P.ld_qkv, 0, P.T);
    sv.load(vb, P.ld_qkv, 0, P.T);
    for (int kt_i = 0; kt_i < ntiles; ++kt_i) {
        __syncthreads();
        sk.store(kt);
        sv.store(vt);
        __syncthreads();
        if (kt_i + 1 < ntiles) {
            sk.load(kb, P.ld_qkv, (kt_i + 1) * BT, P.T);
            sv.load(vb, P.ld_qkv, (kt_i + 1) * BT, P.T);
        }
        f32x4 s[4];
        tile_times_rows<T>(kt, qf, s, lane);          // s[i][r] = raw S^T[key 16i+4g+r][query li]
        const bool edge = km != nullptr || (kt_i + 1) * BT > P.T;      // wave-uniform: masks / tail only here
        float mt;
        if (!edge) {
            mt = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
#pragma unroll
            for (int i = 1; i < 4; ++i) mt = fmaxf(fmaxf(mt, s[i][0]), fmaxf(fmaxf(s[i][1], s[i][2]), s[i][3]));
            mt *= P.scale;
        } else {
            const int key0 = kt_i * BT + 4 * g;
            mt = -INFINITY;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = key0 + 16 * i + r;
                    float x = s[i][r] * P.scale;
                    if (km && key < P.T && !km[key]) x = MASKED_LOG2;
                    if (key >= P.T) x = -INFINITY;
                    s[i][r] = x;                     // already scaled on this path
                    mt = fmaxf(mt, x);
                }
        }
        mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        if (__any(mt > m + P.lazy)) {                 // first tile: m = -inf -> always taken, alpha = 0
            const float mn = fmaxf(m, mt);
            const float alpha = fast_exp2(m - mn);
            m = mn;
            l *= alpha;
#pragma unroll
            for (int db = 0; db < 4; ++db) oacc[db] *= alpha;
        }
        float ls = 0.f;
        if (!edge) {
            const float nb = -m;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[i][r] = fast_exp2(__builtin_fmaf(s[i][r], P.scale, nb)); ls += s[i][r]; }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[i][r] = fast_exp2(s[i][r] - m); ls += s[i][r]; }
        }
        l += ls;
        if (P.thr) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t h0 = hfin(hb + (uint32_t)(8 * i) * HC1), h1 = hfin(hb + (uint32_t)(8 * i + 1) * HC1);
                s[i][0] = (h0 & 0xFFFFu) >= P.thr ? s[i][0] : 0.f;
                s[i][1] = (h0 >> 16) >= P.thr ? s[i][1] : 0.f;
                s[i][2] = (h1 & 0xFFFFu) >= P.thr ? s[i][2] : 0.f;
                s[i][3] = (h1 >> 16) >= P.thr ? s[i][3] : 0.f;
            }
            hb += (uint32_t)(BT / 2) * HC1;
        }
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            oacc[db] = ATile<T>::mma_xt_p(vt, 0, 16 * db, s[0], s[1], oacc[db], lane);
            oacc[db] = ATile<T>::mma_xt_p(vt, 32, 16 * db, s[2], s[3], oacc[db], lane);
        }
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (qi < P.T) {
        const float inv = P.pscale / l;               // 1/(1-p) of the dropout is applied here, once
        T* op = reinterpret_cast<T*>(P.out) + ((long long)b * P.T + qi) * P.ld_o + h * DH + 4 * g;
#pragma unroll
        for (int db = 0; db < 4; ++db) store4<T>(op + 16 * db, oacc[db][0] * inv, oacc[db][1] * inv, oacc[db][2] * inv, oacc[db][3] * inv);
        if (g == 0) P.lse[((long long)bh) * P.T + qi] = m + log2f(l);     // log2-domain log-sum-exp
    }
}

// ================================================================== delta = rowsum(dO * O)
template <typename T>
__global__ void attn_delta_kernel(const AttnP P) {
    const long long n = (long long)P.B * P.T * P.H;
    const long long idx = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 3;     // 8 lanes per (b,t,h)
    const int sub = threadIdx.x & 7;
    float s = 0.f;
    if (idx < n) {
        const int h = (int)(idx % P.H);
        const long long bt = idx / P.H;
        const T* op = reinterpret_cast<const T*>(P.o) + bt * P.ld_o + h * DH + sub * 8;
        const T* dp = reinterpret_cast<const T*>(P.dout) + bt * P.ld_o + h * DH + sub * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += to_f(op[i]) * to_f(dp[i]);
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (idx < n && sub == 0) {
        const int h = (int)(idx % P.H);
        const long long bt = idx / P.H;
        const int b = (int)(bt / P.T), t = (int)(bt % P.T);
        P.delta[((long long)(b * P.H + h)) * P.T + t] = s;
    }
}

// ================================================================== backward A: dK, dV (key block stationary)
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(const AttnP P) {
    __shared__ __attribute__((aligned(16))) char lds[2 * ATile<T>::BYTES];
    __shared__ __attribute__((aligned(16))) float s_lse[BT], s_delta[BT];
    char* qt = lds;
    char* dot = lds + ATile<T>::BYTES;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
    int blk, bh;
    block_coords((P.T + BT - 1) / BT, P.B * P.H, blk, bh);
    const int b = bh / P.H, h = bh % P.H;
    const int key = blk * BT + wv * 16 + li;
    const T* qb = reinterpret_cast<const T*>(P.q) + (long long)b * P.T * P.ld_qkv + h * DH;
    const T* kb = reinterpret_cast<const T*>(P.k) + (long long)b * P.T * P.ld_qkv + h * DH;
    const T* vb = reinterpret_cast<const T*>(P.v) + (long long)b * P.T * P.ld_qkv + h * DH;
    const T* dob = reinterpret_cast<const T*>(P.dout) + (long long)b * P.T * P.ld_o + h * DH;
    RowFrags<T> kf, vf;
    kf.load(kb + (long long)key * P.ld_qkv, key < P.T, lane);
    vf.load(vb + (long long)key * P.ld_qkv, key < P.T, lane);
    const bool key_ok = key < P.T;
    const bool key_masked = P.keymask && key_ok && !P.keymask[(long long)b * P.T + key];

    f32x4 dk[4], dv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { dk[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const int ntiles = (P.T + BT - 1) / BT;
    // dropout word of (query qq, this lane's key): fin(hk + (qq - tile base - 4g) * ThC); field (key & 1)
    const uint32_t ThC = (uint32_t)P.Th * HC1;
    uint32_t hk = (pair_base(bh, P.T, P.Th, 4 * g) + (uint32_t)(key >> 1) + P.seed) * HC1;
    const uint32_t fsh = (key & 1) << 4;
    AStage<T> sq, sd;
    sq.load(qb, P.ld_qkv, 0, P.T);
    sd.load(dob, P.ld_o, 0, P.T);
    for (int qt_i = 0; qt_i < ntiles; ++qt_i) {
        __syncthreads();
        sq.store(qt);
        sd.store(dot);
        if (threadIdx.x < BT) {                        // rows beyond T: Q = dO = 0 (staged zero), lse = delta = 0 -> dS = 0, P^T dO = 0
            const int qq = qt_i * BT + threadIdx.x;
            s_lse[threadIdx.x] = qq < P.T ? P.lse[(long long)bh * P.T + qq] : 0.f;
            s_delta[threadIdx.x] = qq < P.T ? P.delta[(long long)bh * P.T + qq] : 0.f;
        }
        __syncthreads();
        if (qt_i + 1 < ntiles) {
            sq.load(qb, P.ld_qkv, (qt_i + 1) * BT, P.T);
            sd.load(dob, P.ld_o, (qt_i + 1) * BT, P.T);
        }
        f32x4 s[4], dp[4];
        rows_times_tile<T>(qt, kf, s, lane);           // s[i][r]  = raw S[q 16i+4g+r][key li]
        rows_times_tile<T>(dot, vf, dp, lane);         // dp[i][r] = (dO V^T)[q][key]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 lse4 = *reinterpret_cast<const f32x4*>(&s_lse[16 * i + 4 * g]);
            const f32x4 del4 = *reinterpret_cast<const f32x4*>(&s_delta[16 * i + 4 * g]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = __builtin_fmaf(s[i][r], P.scale, -lse4[r]);
                if (P.keymask && key_masked) x = MASKED_LOG2 - lse4[r];
                const float p = fast_exp2(x);          // keys beyond T: harmless garbage in rows that are never stored
                float pd = p, dpv = dp[i][r];
                if (P.thr) {
                    const uint32_t hh = hfin(hk + (uint32_t)(16 * i + r) * ThC);
                    const bool keep = ((hh >> fsh) & 0xFFFFu) >= P.thr;
                    pd = keep ? p : 0.f;
                    dpv = keep ? dpv : 0.f;
                }
                s[i][r] = pd;                                                     // P (dropped, unscaled) for dV
                dp[i][r] = p * __builtin_fmaf(dpv, P.pscale, -del4[r]);          // dS / softmax scale for dK
            }
        }
        hk += (uint32_t)BT * ThC;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            dv[db] = ATile<T>::mma_xt_p(dot, 0, 16 * db, s[0], s[1], dv[db], lane);
            dv[db] = ATile<T>::mma_xt_p(dot, 32, 16 * db, s[2], s[3], dv[db], lane);
            dk[db] = ATile<T>::mma_xt_p(qt, 0, 16 * db, dp[0], dp[1], dk[db], lane);
            dk[db] = ATile<T>::mma_xt_p(qt, 32, 16 * db, dp[2], dp[3], dk[db], lane);
        }
    }
    if (key_ok) {
        T* dkp = reinterpret_cast<T*>(P.dk) + ((long long)b * P.T + key) * P.ld_d + h * DH + 4 * g;
        T* dvp = reinterpret_cast<T*>(P.dv) + ((long long)b * P.T + key) * P.ld_d + h * DH + 4 * g;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            store4<T>(dkp + 16 * db, dk[db][0] * P.dscale, dk[db][1] * P.dscale, dk[db][2] * P.dscale, dk[db][3] * P.dscale);
            store4<T>(dvp + 16 * db, dv[db][0] * P.pscale, dv[db][1] * P.pscale, dv[db][2] * P.pscale, dv[db][3] * P.pscale);
        }
    }
}

// ================================================================== backward B: dQ (query block stationary)
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(const AttnP P) {
    __shared__ __attribute__((aligned(16))) char lds[2 * ATile<T>::BYTES];
    char* kt = lds;
    char* vt = lds + ATile<T>::BYTES;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
    int blk, bh;
    block_coords((P.T + BT - 1) / BT, P.B * P.H, blk, bh);
    const int b = bh / P.H, h = bh % P.H;
    const int qi = blk * BT + wv * 16 + li;
    const bool q_ok = qi < P.T;
    const T* qb = reinterpret_cast<const T*>(P.q) + (long long)b * P.T * P.ld_qkv + h * DH;
    const T* kb = reinterpret_cast<const T*>(P.k) + (long long)b * P.T * P.ld_qkv + h * DH;
    const T* vb = reinterpret_cast<const T*>(P.v) + (long long)b * P.T * P.ld_qkv + h * DH;
    const T* dob = reinterpret_cast<const T*>(P.dout) + (long long)b * P.T * P.ld_o + h * DH;
    RowFrags<T> qf, dof;
    qf.load(qb + (long long)qi * P.ld_qkv, q_ok, lane);
    dof.load(dob + (long long)qi * P.ld_o, q_ok, lane);
    const float lse = q_ok ? P.lse[(long long)bh * P.T + qi] : 0.f;
    const float delta = q_ok ? P.delta[(long long)bh * P.T + qi] : 0.f;
    const uint8_t* km = P.keymask ? P.keymask + (long long)b * P.T : nullptr;

    f32x4 dq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dq[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int ntiles = (P.T + BT - 1) / BT;
    uint32_t hb = (pair_base(bh, P.T, P.Th, qi) + (uint32_t)(2 * g) + P.seed) * HC1;
    const float nlse = -lse, ndelta = -delta;
    AStage<T> sk, sv;
    sk.load(kb, P.ld_qkv, 0, P.T);
    sv.load(vb, P.ld_qkv, 0, P.T);
    for (int kt_i = 0; kt_i < ntiles; ++kt_i) {
        __syncthreads();
        sk.store(kt);
        sv.store(vt);
        __syncthreads();
        if (kt_i + 1 < ntiles) {
            sk.load(kb, P.ld_qkv, (kt_i + 1) * BT, P.T);
            sv.load(vb, P.ld_qkv, (kt_i + 1) * BT, P.T);
        }
        f32x4 s[4], dp[4];
        tile_times_rows<T>(kt, qf, s, lane);           // raw S^T[key][q]
        tile_times_rows<T>(vt, dof, dp, lane);         // (V dO^T)[key][q]
        // keys beyond T: staged K = V = 0, so dS is finite garbage that multiplies zero K rows -- no tail test needed
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t h0 = 0, h1 = 0;
            if (P.thr) { h0 = hfin(hb + (uint32_t)(8 * i) * HC1); h1 = hfin(hb + (uint32_t)(8 * i + 1) * HC1); }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = __builtin_fmaf(s[i][r], P.scale, nlse);
                if (km) {                                                  // wave-uniform
                    const int key = kt_i * BT + 4 * g + 16 * i + r;
                    if (key < P.T && !km[key]) x = MASKED_LOG2 + nlse;
                }
                const float p = fast_exp2(x);
                float dpv = dp[i][r];
                if (P.thr) {
                    const uint32_t hh = (r < 2) ? h0 : h1;
                    dpv = (((r & 1) ? (hh >> 16) : (hh & 0xFFFFu)) >= P.thr) ? dpv : 0.f;
                }
                s[i][r] = p * __builtin_fmaf(dpv, P.pscale, ndelta);      // dS^T[key][q] / softmax scale
            }
        }
        hb += (uint32_t)(BT / 2) * HC1;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            dq[db] = ATile<T>::mma_xt_p(kt, 0, 16 * db, s[0], s[1], dq[db], lane);
            dq[db] = ATile<T>::mma_xt_p(kt, 32, 16 * db, s[2], s[3], dq[db], lane);
        }
    }
    if (q_ok) {
        T* dqp = reinterpret_cast<T*>(P.dq) + ((long long)b * P.T + qi) * P.ld_d + h * DH + 4 * g;
#pragma unroll
        for (int db = 0; db < 4; ++db) store4<T>(dqp + 16 * db, dq[db][0] * P.dscale, dq[db][1] * P.dscale, dq[db][2] * P.dscale, dq[db][3] * P.dscale);
    }
}

void set_softmax_params(AttnP& P, float scale, float p, uint32_t seed) {
    P.scale = scale * 1.4426950408889634f;
    P.dscale = scale;
    P.lazy = 8.0f;
    P.Th = (P.T + 1) / 2;
    uint32_t thr = p > 0.f ? (uint32_t)(p * 65536.0f + 0.5f) : 0u;
    if (p > 0.f && thr == 0) thr = 1;
    P.thr = thr;
    P.pscale = thr ? 65536.0f / (65536.0f - (float)thr) : 1.0f;      // unbiased for the realised keep probability
    P.seed = seed;
}

int check_common(const char* what, int dtype, int B, int T_, int H, long long ld_qkv, long long ld_o, float p) {
    A8_REQUIRE(dtype == A8_F32 || dtype == A8_BF16, "%s: bad dtype", what);
    A8_REQUIRE(B >= 1 && T_ >= 1 && H >= 1, "%s: bad sizes B=%d T=%d H=%d", what, B, T_, H);
    const int epc = dtype == A8_BF16 ? 8 : 4;
    A8_REQUIRE(ld_qkv % epc == 0 && ld_o % epc == 0 && ld_qkv >= (long long)H * DH && ld_o >= (long long)H * DH, "%s: bad leading dimensions", what);
    A8_REQUIRE(p >= 0.f && p < 1.f, "%s: bad dropout p", what);
    A8_REQUIRE((long long)B * H * T_ * ((T_ + 1) / 2) < (1ll << 32), "%s: B*H*T*T/2 exceeds the dropout counter range", what);
    return A8_OK;
}

}  // namespace

extern "C" int a8_attn_fwd(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* keymask,
                           int B, int T_, int H, int64_t ld_qkv, int64_t ld_o, float scale, float p, uint32_t seed, void* stream) {
    if (int e = check_common("a8_attn_fwd", dtype, B, T_, H, ld_qkv, ld_o, p)) return e;
    AttnP P = {};
    P.q = (const char*)q; P.k = (const char*)k; P.v = (const char*)v; P.out = (char*)o; P.lse = lse; P.keymask = keymask;
    P.B = B; P.T = T_; P.H = H; P.ld_qkv = ld_qkv; P.ld_o = ld_o;
    set_softmax_params(P, scale, p, seed);
    const dim3 grid(cdiv(T_, BT) * B * H);
    if (dtype == A8_BF16) hipLaunchKernelGGL(attn_fwd_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, P);
    else hipLaunchKernelGGL(attn_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, P);
    return a8_check_launch("a8_attn_fwd");
}

extern "C" int a8_attn_bwd(int dtype, const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                           const uint8_t* keymask, void* dq, void* dk, void* dv, float* delta, int B, int T_, int H, int64_t ld_qkv,
                           int64_t ld_o, int64_t ld_dqkv, float scale, float p, uint32_t seed, void* stream) {
    if (int e = check_common("a8_attn_bwd", dtype, B, T_, H, ld_qkv, ld_o, p)) return e;
    AttnP P = {};
    P.q = (const char*)q; P.k = (const char*)k; P.v = (const char*)v; P.o = (const char*)o; P.dout = (const char*)dout;
    P.lse = (float*)lse; P.delta = delta; P.keymask = keymask;
    P.dq = (char*)dq; P.dk = (char*)dk; P.dv = (char*)dv;
    P.B = B; P.T = T_; P.H = H; P.ld_qkv = ld_qkv; P.ld_o = ld_o; P.ld_d = ld_dqkv;
    set_softmax_params(P, scale, p, seed);
    hipStream_t s = (hipStream_t)stream;
    const long long nrow = (long long)B * T_ * H;
    const dim3 grid(cdiv(T_, BT) * B * H);
    if (dtype == A8_BF16) {
        hipLaunchKernelGGL(attn_delta_kernel<bf16>, dim3(cdiv(nrow * 8, 256)), dim3(256), 0, s, P);
        hipLaunchKernelGGL(attn_bwd_kv_kernel<bf16>, grid, dim3(256), 0, s, P);
        hipLaunchKernelGGL(attn_bwd_q_kernel<bf16>, grid, dim3(256), 0, s, P);
    } else {
        hipLaunchKernelGGL(attn_delta_kernel<float>, dim3(cdiv(nrow * 8, 256)), dim3(256), 0, s, P);
        hipLaunchKernelGGL(attn_bwd_kv_kernel<float>, grid, dim3(256), 0, s, P);
        hipLaunchKernelGGL(attn_bwd_q_kernel<float>, grid, dim3(256), 0, s, P);
    }
    return a8_check_launch("a8_attn_bwd");
}
