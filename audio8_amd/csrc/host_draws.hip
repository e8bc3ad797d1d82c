// a8hip host side: the negative-index draw of the contrastive loss, on the host, bit for bit as numpy's legacy global RNG makes it.
// The reference samples negatives with `np.random.randint(0, T - 1, (B, N * T))` followed by the `>= own -> +1` shift
// (audio8/wav2vec2.py:955-976).  At cfg2 that is 284 000 masked-rejection draws on numpy's MT19937 per step -- 2.4-3.9 ms of host time,
// a fifth of the step's host budget, for a loop that is a few instructions per value.  This file restates numpy's algorithm
// (numpy/random/src/mt19937/mt19937.c, src/distributions/distributions.c: random_bounded_uint64_fill -> the 32-bit masked-rejection
// generator for ranges below 2^32; the legacy RandomState always takes the masked path) on a copy of the generator's state, which the
// caller reads with np.random.get_state() and writes back with set_state(): the global stream continues exactly where numpy would have
// left it.  No GPU work here; it lives in this library because it is host code of the hot path's boundary.
#include "a8_common.h"
#include "a8hip_internal.h"

namespace {

constexpr int MT_N = 624, MT_M = 397;

inline void mt_twist(uint32_t* key) {
    constexpr uint32_t A = 0x9908b0dfu, UP = 0x80000000u, LO = 0x7fffffffu;
    int i = 0;
    uint32_t y;
    for (; i < MT_N - MT_M; ++i) {
        y = (key[i] & UP) | (key[i + 1] & LO);
        key[i] = key[i + MT_M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & A);
    }
    for (; i < MT_N - 1; ++i) {
        y = (key[i] & UP) | (key[i + 1] & LO);
        key[i] = key[i + (MT_M - MT_N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & A);
    }
    y = (key[MT_N - 1] & UP) | (key[0] & LO);
    key[MT_N - 1] = key[MT_M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & A);
}

inline uint32_t mt_next(uint32_t* key, int& pos) {
    if (pos == MT_N) { mt_twist(key); pos = 0; }
    uint32_t y = key[pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

}  // namespace

// out[b * n_neg * T + j] = v + (v >= j / n_neg) + b * T with v the next draw of randint(0, T - 1) -- the flat row indices of the
// negatives of masked frame j / n_neg of utterance b (audio8/wav2vec2.py:962-970).  key[624] / *pos: numpy's MT19937 state, updated.
extern "C" int a8_host_negative_indices(uint32_t* key, int* pos, int B, int T, int n_neg, int32_t* out) {
    A8_REQUIRE(key && pos && out, "a8_host_negative_indices: null argument");
    A8_REQUIRE(*pos >= 0 && *pos <= MT_N, "a8_host_negative_indices: bad generator position %d", *pos);
    A8_REQUIRE(B >= 0 && n_neg >= 1 && T >= 2 && (long long)B * T < (1ll << 31), "a8_host_negative_indices: bad sizes B=%d T=%d N=%d", B, T, n_neg);
    const uint32_t rng = (uint32_t)(T - 2);                 // randint's high is exclusive: values 0 .. T - 2
    int p = *pos;
    if (rng == 0) {                                         // numpy draws nothing for a one-value range
        for (int b = 0; b < B; ++b)
            for (long long j = 0; j < (long long)n_neg * T; ++j) out[(long long)b * n_neg * T + j] = (0 >= j / n_neg ? 1 : 0) + b * T;
        return A8_OK;
    }
    uint32_t mask = rng;                                    // smallest 2^k - 1 >= rng (gen_mask)
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    // pass 1: accepted draws, in order, straight into `out` (branch-free compaction: a third of the draws are rejected at cfg2 and a
    // data-dependent branch per draw costs more than the generator); never more outputs per batch than values still missing, so the
    // generator stops exactly behind the draw that produced the last value, as numpy's loop does
    const long long total = (long long)B * n_neg * T;
    long long n = 0;
    while (n < total) {
        if (p == MT_N) { mt_twist(key); p = 0; }
        const long long room = total - n;
        const int cnt = (int)(room < (long long)(MT_N - p) ? room : (long long)(MT_N - p));
        for (int i = 0; i < cnt; ++i) {
            uint32_t y = key[p + i];
            y ^= y >> 11;
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= y >> 18;
            const uint32_t v = y & mask;
            out[n] = (int32_t)v;
            n += v <= rng ? 1 : 0;
        }
        p += cnt;
    }
    // pass 2: the reference's shift past the frame's own index and the utterance's row offset
    for (int b = 0; b < B; ++b) {
        int32_t* row = out + (long long)b * n_neg * T;
        const int32_t off = b * T;
        for (int t = 0; t < T; ++t)
            for (int k = 0; k < n_neg; ++k) {
                const int32_t v = row[(long long)t * n_neg + k];
                row[(long long)t * n_neg + k] = v + (v >= t ? 1 : 0) + off;
            }
    }
    *pos = p;
    return A8_OK;
}
