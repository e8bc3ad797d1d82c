// Archived experiment (round 3): pipelined LayerNorm backward for 16-bit rows of NC * 256 columns -- 4 columns per lane per 256-column
// group, the next row's operands requested before the current row is reduced, one LDS pass for the three column sums, 6-wave blocks
// x 3 rows.  Measured at cfg2 (6248 x 768, tools/ln_ab.py): 24.0 / 25.5 us (without / with the second gradient, cold operands)
// against 25.1 / 27.5 us for the shipped kernel; in the step (tools/step_ab.py) no difference (12.27 vs 12.18 ms, noise 0.1).  The
// kernel is not bound by its per-row latency chain: 54 MB in ~18 us plus ~6 us of column-sum flush is what this access pattern gets.
// Not compiled; kept for the record (fragments of csrc/norm_elem.hip at that commit).
#if 0
// ------------------------------------------------------------------ LayerNorm backward, 16-bit, cols = NC * 256 (round 3)
// The kernel above is bound by latency, not bandwidth (28 us for 54 MB at cfg2): a wave walks its rows one after the other, each row
// a load -> two reductions -> store chain, and with 1.5 sixteen-byte chunks per lane a quarter of the lanes idle.  Here a lane owns 4
// consecutive columns of each 256-column group (NC loads of 8 bytes per operand, every lane busy), the NEXT row's operands are
// requested before the current row is reduced (two register sets, loop unrolled by two), the wave sums are DPP / permlane (no LDS
// crossbar), dropout / second-gradient presence are template flags, and the three column sums leave the block through ONE LDS pass
// (4 slabs per sum, upper waves add into the lower waves' slabs) instead of three.
struct LnRaw3 { uint2 x, r, g, g2; };
A8_DEV void unpack4(uint2 v, float (&f)[4]) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xFFFF0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xFFFF0000u);
}
A8_DEV uint2 pack4(const float (&f)[4]) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t a = {(bf16)f[0], (bf16)f[1]}, b = {(bf16)f[2], (bf16)f[3]};
    uint2 o;
    o.x = *reinterpret_cast<const uint32_t*>(&a);
    o.y = *reinterpret_cast<const uint32_t*>(&b);
    return o;
}
template <int NW, int NC, bool DY2, bool DROP>
__global__ __launch_bounds__(NW * 64, NC <= 3 ? 3 : 2) void ln_bwd4_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ dy2, const bf16* __restrict__ x,
                                                              const bf16* __restrict__ res, const float* __restrict__ gamma,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              bf16* __restrict__ dx, bf16* __restrict__ dres, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float* __restrict__ dbias, int rows, Drop dr, long long rng0) {
    constexpr int COLS = NC * 256, NS = 4;
    __shared__ __attribute__((aligned(16))) float lds[3 * NS * COLS];
    __shared__ __attribute__((aligned(16))) float sgam[COLS];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float inv = 1.0f / (float)COLS;
    float ag[NC][4], ab[NC][4], ad[NC][4];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) { ag[c][i] = 0.f; ab[c][i] = 0.f; ad[c][i] = 0.f; }
    for (int i = threadIdx.x; i < COLS; i += NW * 64) sgam[i] = gamma[i];
    __syncthreads();
    const int stride = gridDim.x * NW;
    auto request = [&](LnRaw3 (&raw)[NC], float& mu, float& rs, int row) {
        const long long base = (long long)row * COLS + lane * 4;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            raw[c].x = *reinterpret_cast<const uint2*>(x + base + c * 256);
            raw[c].r = *reinterpret_cast<const uint2*>(res + base + c * 256);
            raw[c].g = *reinterpret_cast<const uint2*>(dy + base + c * 256);
            if (DY2) raw[c].g2 = *reinterpret_cast<const uint2*>(dy2 + base + c * 256);
        }
        mu = mean[row]; rs = rstd[row];
    };
    auto process = [&](const LnRaw3 (&raw)[NC], float mu, float rs, int row) {
        const long long base = (long long)row * COLS + lane * 4;
        float xh[NC][4], g[NC][4];
        uint32_t keep[NC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float v[4], r[4];
            unpack4(raw[c].x, v);
            unpack4(raw[c].r, r);
            unpack4(raw[c].g, g[c]);
            keep[c] = 0u;
            if (DROP) {
                // element e of the 4 is kept when bit 8 e + 7 is set: the word of 8-chunk (off >> 3), half (off >> 2) & 1 (a8_keep8)
                keep[c] = ((a8_hash(dr.seed, dr.stream, (uint32_t)((base + c * 256 + rng0) >> 2)) & 0x7F7F7F7Fu) | 0x80808080u) - dr.thr;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = (keep[c] >> (8 * i + 7)) & 1u ? v[i] * dr.scale : 0.f;
            }
            if (DY2) {
                float r2[4];
                unpack4(raw[c].g2, r2);
#pragma unroll
                for (int i = 0; i < 4; ++i) g[c][i] += r2[i];
            }
            const f32x4 gm = *reinterpret_cast<const f32x4*>(sgam + c * 256 + lane * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xh[c][i] = (v[i] + r[i] - mu) * rs;
                ag[c][i] += g[c][i] * xh[c][i];
                ab[c][i] += g[c][i];
                g[c][i] *= gm[i];
                s1 += g[c][i];
                s2 += g[c][i] * xh[c][i];
            }
        }
        const float a = wave_sum(s1) * inv, b = wave_sum(s2) * inv;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = rs * (g[c][i] - a - xh[c][i] * b);
            if (dres) *reinterpret_cast<uint2*>(dres + base + c * 256) = pack4(o);
            if (DROP) {
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (keep[c] >> (8 * i + 7)) & 1u ? o[i] * dr.scale : 0.f;
            }
            *reinterpret_cast<uint2*>(dx + base + c * 256) = pack4(o);
#pragma unroll
            for (int i = 0; i < 4; ++i) ad[c][i] += o[i];
        }
    };
    {
        LnRaw3 ra[NC], rb[NC];
        float mua = 0.f, rsa = 0.f, mub = 0.f, rsb = 0.f;
        int row = blockIdx.x * NW + w;
        if (row < rows) request(ra, mua, rsa, row);
        while (row < rows) {
            const int r1 = row + stride, r2 = r1 + stride;
            if (r1 < rows) request(rb, mub, rsb, r1);
            process(ra, mua, rsa, row);
            if (r1 >= rows) break;
            if (r2 < rows) request(ra, mua, rsa, r2);
            process(rb, mub, rsb, r1);
            row = r2;
        }
    }
    // column sums: waves 0..3 store their registers into slab w of each sum, waves 4.. add into slab w - 4, then one atomic per column
    float* sg = lds, *sb = lds + NS * COLS, *sd = lds + 2 * NS * COLS;
    for (int pass = 0; pass < (NW + NS - 1) / NS; ++pass) {
        if (w / NS == pass) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int o = (w % NS) * COLS + c * 256 + lane * 4;
                f32x4 tg = {ag[c][0], ag[c][1], ag[c][2], ag[c][3]}, tb = {ab[c][0], ab[c][1], ab[c][2], ab[c][3]},
                      td = {ad[c][0], ad[c][1], ad[c][2], ad[c][3]};
                if (pass > 0) {
                    tg += *reinterpret_cast<const f32x4*>(sg + o);
                    tb += *reinterpret_cast<const f32x4*>(sb + o);
                    td += *reinterpret_cast<const f32x4*>(sd + o);
                }
                *reinterpret_cast<f32x4*>(sg + o) = tg;
                *reinterpret_cast<f32x4*>(sb + o) = tb;
                *reinterpret_cast<f32x4*>(sd + o) = td;
            }
        }
        __syncthreads();
    }
    constexpr int NSU = NW < NS ? NW : NS;
    for (int i = threadIdx.x; i < 3 * COLS; i += NW * 64) {
        const int which = i / COLS, col = i - which * COLS;
        const float* sl = lds + which * NS * COLS + col;
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < NSU; ++k) t += sl[k * COLS];
        float* out = which == 0 ? dgamma : which == 1 ? dbeta : dbias;
        if (out) atomicAdd(out + col, t);
    }
}


// ---- launcher fragment
    if (dtype == A8_BF16 && g_a8_tuning[7] == 0 && res && dx && dgamma && (cols == 512 || cols == 768 || cols == 1024) && rng_elem0 % 4 == 0) {
        // 6 waves x R rows: two blocks per CU at 3 waves per SIMD (a 168-register budget holds the two operand sets), R = 3 puts
        // cfg2's 6248 rows into 348 blocks -- one round of the chip's 512 slots and fewer column-sum atomics than 391 blocks of 16 rows
        const int r4 = g_a8_tuning[8] > 0 ? g_a8_tuning[8] : 3, grid4 = cdiv(rows, 6 * r4);
        // (a8_set_tuning key 7 = 1 keeps the general kernel below, for A/B runs; it also serves f32, other widths and optional operands)
#define LN_BWD4(NC) do {                                                                                                                   \
        const int which = (dy2 ? 2 : 0) + (dr.thr ? 1 : 0);                                                                              \
        auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(grid4), dim3(384), 0, (hipStream_t)stream, (const bf16*)dy, (const bf16*)dy2, \
                                                      (const bf16*)x, (const bf16*)res, gamma, mean, rstd, (bf16*)dx, (bf16*)dres, dgamma, dbeta,  \
                                                      dbias, rows, dr, (long long)rng_elem0); };                                           \
        if (which == 0) go(ln_bwd4_kernel<6, NC, false, false>); else if (which == 1) go(ln_bwd4_kernel<6, NC, false, true>);            \
        else if (which == 2) go(ln_bwd4_kernel<6, NC, true, false>); else go(ln_bwd4_kernel<6, NC, true, true>);                         \
    } while (0)
        if (cols == 512) LN_BWD4(2); else if (cols == 768) LN_BWD4(3); else LN_BWD4(4);
#undef LN_BWD4
        return a8_check_launch("a8_layernorm_bwd");
    }
#endif
