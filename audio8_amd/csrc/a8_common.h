// a8hip -- common device/host helpers for the MI355X (gfx950) wav2vec2 hot path.
// Wave = 64 lanes everywhere.  All MFMA code is written once against Mma<T> for
// T = bf16 (v_mfma_f32_16x16x32_bf16) and T = float (v_mfma_f32_16x16x4_f32, exact f32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

typedef __bf16 bf16;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define A8_DEV __device__ __forceinline__
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

// ---------------------------------------------------------------- status / errors (host)
enum { A8_OK = 0, A8_ERR_ARG = 1, A8_ERR_LAUNCH = 2, A8_ERR_UNSUPPORTED = 3 };
enum { A8_F32 = 0, A8_BF16 = 1 };
int a8_set_error(int code, const char* fmt, ...);
int a8_check_launch(const char* what);
#define A8_REQUIRE(cond, ...) do { if (!(cond)) return a8_set_error(A8_ERR_ARG, __VA_ARGS__); } while (0)

// ---------------------------------------------------------------- small math
// Exact (erf) GELU without erff: Phi(x) = 0.5 erfc(-x / sqrt 2), with log2 erfc(|x| / sqrt 2) as a degree-7
// polynomial (weighted least-squares fit on [0, 5.75], exact 0 at 0) -- branch-free, one v_exp_f32.
// Max abs error vs float64 over [-8, 8]: Phi 1.2e-7, gelu 7.7e-7 (torch's own f32 GELU: 1.4e-6), gelu' 1.7e-7.
A8_DEV float norm_cdf_f(float x) {
    const float t = fminf(fabsf(x), 5.75f);
    float r = 8.510881681e-06f;
    r = fmaf(r, t, -5.449422293e-05f);
    r = fmaf(r, t, -4.180608761e-04f);
    r = fmaf(r, t, 7.381212995e-03f);
    r = fmaf(r, t, -5.268092407e-02f);
    r = fmaf(r, t, -4.591593132e-01f);
    r = fmaf(r, t, -1.151109407e+00f);
    const float e = __builtin_amdgcn_exp2f(r * t);       // erfc(|x| / sqrt 2) in (8e-9, 1]
    return 0.5f + copysignf(fmaf(-0.5f, e, 0.5f), x);
}
A8_DEV float gelu_f(float x) { return x * norm_cdf_f(x); }
A8_DEV float gelu_grad_f(float x) {
    // d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
    const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);
    return fmaf(x, pdf, norm_cdf_f(x));
}
// The same functions two elements at a time: the polynomial and the products become v_pk_fma_f32 / v_pk_mul_f32 (two f32 lanes per
// instruction at the single-lane issue cost; the scalar form's v_fmaak literals cannot pack): 9.3 instead of 13.4 vector instructions per
// GELU, the same IEEE operations in the same order -- bit-identical results.  Epilogues and element-wise kernels call the array forms.
typedef float f32x2 __attribute__((ext_vector_type(2)));
A8_DEV f32x2 splat2(float c) { return (f32x2){c, c}; }
A8_DEV f32x2 norm_cdf2(f32x2 x) {
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
A8_DEV f32x2 gelu_grad2(f32x2 x) {
    const f32x2 a = (x * splat2(-0.72134752044448170368f)) * x;
    const f32x2 pdf = splat2(0.39894228040143267794f) * (f32x2){__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
    return __builtin_elementwise_fma(x, pdf, norm_cdf2(x));
}
template <int N> A8_DEV void gelu_n(float (&v)[N]) {                      // v = gelu(v)
    static_assert(N % 2 == 0, "pairs");
#pragma unroll
    for (int i = 0; i < N; i += 2) {
        const f32x2 x = {v[i], v[i + 1]}, y = x * norm_cdf2(x);
        v[i] = y[0]; v[i + 1] = y[1];
    }
}
template <int N> A8_DEV void mul_gelu_grad_n(float (&v)[N], const float (&z)[N]) {      // v *= gelu'(z)
    static_assert(N % 2 == 0, "pairs");
#pragma unroll
    for (int i = 0; i < N; i += 2) {
        const f32x2 g = gelu_grad2((f32x2){z[i], z[i + 1]});
        v[i] *= g[0]; v[i + 1] *= g[1];
    }
}

template <typename T> A8_DEV float to_f(T v) { return (float)v; }
template <typename T> A8_DEV T from_f(float v) { return (T)v; }

// Reductions over the 4 rows of 16 lanes (lanes with equal lane&15), all-VALU via the gfx950 permlane swaps
// (no ds_bpermute round trip through the LDS pipeline).  Every lane receives the result.
A8_DEV float rows_max(float x) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    float lo = __uint_as_float(a[0]), hi = __uint_as_float(a[1]), m;
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(lo), "v"(hi));
    auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    lo = __uint_as_float(b[0]); hi = __uint_as_float(b[1]);
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(lo), "v"(hi));
    return m;
}
A8_DEV float rows_sum(float x) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    float m = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// Whole-wave reductions, all-VALU: four DPP steps inside each row of 16 lanes (quad xor 1, quad xor 2, half-row mirror, row mirror),
// then the two permlane swaps across the rows -- no ds_bpermute (six dependent LDS-crossbar round trips per value before).
template <int CTRL> A8_DEV float dpp_f(float v) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xF, 0xF, true));
}
A8_DEV float wave_sum(float v) {
    v += dpp_f<0xB1>(v);          // quad_perm:[1,0,3,2]
    v += dpp_f<0x4E>(v);          // quad_perm:[2,3,0,1]
    v += dpp_f<0x141>(v);         // row_half_mirror
    v += dpp_f<0x140>(v);         // row_mirror
    return rows_sum(v);
}
A8_DEV float wave_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    v = fmaxf(v, dpp_f<0x140>(v));
    return rows_max(v);
}
// block-wide sum for blockDim.x = NW*64 threads; `red` is NW floats of LDS scratch.
template <int NW> A8_DEV float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += red[i];
    return s;
}

// ---------------------------------------------------------------- counter-based RNG (dropout)
// 32-bit mix of (seed, stream, index): the same function regenerates a mask in backward.
A8_DEV uint32_t a8_hash(uint32_t seed, uint32_t stream, uint32_t idx) {
    uint32_t h = seed ^ (stream * 0x9E3779B9u) ^ (idx * 0x85EBCA6Bu);
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    h += idx; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
// Attention-probability dropout: one cheap 32-bit mix per PAIR of adjacent keys, a 16-bit field per key.
// keep(q, key) = field(key & 1) of mix(seed, pair_base(q) + (key >> 1)) >= thr16, thr16 = round(p * 65536).
A8_DEV uint32_t a8_mix(uint32_t seed, uint32_t idx) {
    uint32_t h = (idx ^ seed) * 0x9E3779B1u;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    return h;
}

// Activation dropout (LayerNorm inputs, feature / input dropout): the keep decisions of the 8 consecutive elements of 16-byte chunk
// `c8` come from TWO hash words, a 7-bit field per element (the drop probability is realised in 1/128 steps, thr7 = round(128 p), and the
// survivors are scaled by 128 / (128 - thr7): unbiased for the realised rate) -- a quarter of the hash work of one word per element.
// Bit 8 e + 7 of the returned word t[e >> 2] is 1 when element e is KEPT: (field | 0x80) - thr7 borrows out of bit 7 exactly when field < thr7.
struct Keep8 { uint32_t t0, t1; };
A8_DEV Keep8 a8_keep8(uint32_t seed, uint32_t stream, uint32_t c8, uint32_t thr7x4 /* thr7 * 0x01010101 */) {
    Keep8 k;
    k.t0 = ((a8_hash(seed, stream, 2u * c8) & 0x7F7F7F7Fu) | 0x80808080u) - thr7x4;
    k.t1 = ((a8_hash(seed, stream, 2u * c8 + 1u) & 0x7F7F7F7Fu) | 0x80808080u) - thr7x4;
    return k;
}
A8_DEV bool a8_kept(const Keep8& k, int e) { return (((e < 4 ? k.t0 : k.t1) >> (8 * (e & 3) + 7)) & 1u) != 0u; }

// ---------------------------------------------------------------- vector load/store of 16 bytes
template <typename T> struct Vec16;                 // 16-byte chunk of T
template <> struct Vec16<bf16> { static constexpr int N = 8; };
template <> struct Vec16<float> { static constexpr int N = 4; };

A8_DEV u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
A8_DEV void st16(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }

// ---------------------------------------------------------------- MFMA abstraction
// D[i][j] += sum_k A[i][k] * B[k][j] on a 16x16 tile.
//   a-fragment: lane l holds A[i = l&15][k-slots of group g = l>>4]
//   b-fragment: lane l holds B[k-slots of group g][j = l&15]
//   accumulator: lane l, reg r holds D[i = 4*g + r][j = l&15]
// bf16: 8 k-slots per lane per instruction (K = 32); f32: 1 (K = 4).
template <typename T> struct Mma;
template <> struct Mma<bf16> {
    static constexpr int KI = 32, KPL = 8;
    typedef bf16x8 Frag;
    static A8_DEV f32x4 mma(Frag a, Frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Mma<float> {
    static constexpr int KI = 4, KPL = 1;
    typedef float Frag;
    static A8_DEV f32x4 mma(Frag a, Frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
};

// ---------------------------------------------------------------- LDS tile images
// (1) "KC" tile: [ROWS][BK] with the reduction index contiguous.
//     bf16: BK = 64 -> 128-byte rows, 16-byte chunk c stored at c ^ (row & 7)  (conflict-free b128 reads)
//     f32 : BK = 16 -> rows of 20 floats (80 B)
// (2) "KS" tile: [BK][128] with the row index contiguous (operand is K-strided in global memory).
//     bf16: 256-byte k-rows, 32-byte chunk c stored at c ^ key(k); read with ds_read_b64_tr_b16
//     f32 : k-rows of 144 floats
template <typename T> struct Tile;
template <> struct Tile<bf16> {
    static constexpr int BK = 64;
    static constexpr int KC_ROW_BYTES = 128;
    static constexpr int KS_ROW_BYTES = 256;
    static A8_DEV int kc_off(int row, int chunk16) { return row * 128 + ((chunk16 ^ (row & 7)) << 4); }
    static A8_DEV int ks_key(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }
    static A8_DEV int ks_off(int k, int piece16) { return k * 256 + ((((piece16 >> 1) ^ ks_key(k))) << 5) + ((piece16 & 1) << 4); }
    // fragment of 16 rows starting at rb for the 32-wide k-step ks of a KC tile
    static A8_DEV bf16x8 kc_frag(const char* tile, int rb, int ks, int lane) {
        const int row = rb + (lane & 15);
        return *reinterpret_cast<const bf16x8*>(tile + kc_off(row, ks * 4 + (lane >> 4)));
    }
    // same fragment from a KS tile (k-slot j of group g = k 8g + j), via two transposed reads
    static A8_DEV bf16x8 ks_frag(const char* tile, int rb, int ks, int lane) {
        const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
        const int nb = rb >> 4;
        const int k0 = ks * 32 + 8 * g + q, k1 = k0 + 4;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + k0 * 256 + ((nb ^ ks_key(k0)) << 5) + 8 * p));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + k1 * 256 + ((nb ^ ks_key(k1)) << 5) + 8 * p));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    }
};
template <> struct Tile<float> {
    static constexpr int BK = 16;
    static constexpr int KC_ROW_BYTES = 80;
    static constexpr int KS_ROW_BYTES = 576;
    static A8_DEV int kc_off(int row, int chunk16) { return row * 80 + (chunk16 << 4); }
    static A8_DEV int ks_off(int k, int piece16) { return k * 576 + (piece16 << 4); }
    static A8_DEV float kc_frag(const char* tile, int rb, int ks, int lane) {
        return *reinterpret_cast<const float*>(tile + (rb + (lane & 15)) * 80 + (ks * 4 + (lane >> 4)) * 4);
    }
    static A8_DEV float ks_frag(const char* tile, int rb, int ks, int lane) {
        return *reinterpret_cast<const float*>(tile + (ks * 4 + (lane >> 4)) * 576 + (rb + (lane & 15)) * 4);
    }
};

// XCD-aware bijective remap of a 1-D block id: consecutive logical ids land on one XCD
// (blocks b and b+8 share an XCD under round-robin dispatch; speed only, never correctness).
A8_DEV int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
