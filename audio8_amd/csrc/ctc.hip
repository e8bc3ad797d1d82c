// a8hip: CTC loss forward + gradient (F.ctc_loss native path with cuDNN disabled, blank = Offsets.GO = 0,
// zero_infinity -- audio8/ctc.py:186-206).
//
// The alpha / beta recurrences over the 2S+1 extended labels are serial in time and tiny per step: the loss is LATENCY, and it
// sits between the model's forward and its backward with nothing to overlap it (round 5: one wave per utterance running alpha,
// then beta + gradient, with a global-memory round trip per frame in each: 6.5 ms of the 21.4 ms cfg5 step).  Round 6:
//   * alpha and beta run CONCURRENTLY, one block each per utterance (grid B x 2), and only recurse; both store their per-frame
//     normalised log2 values; the posteriors and the gradient are a third, fully parallel launch (one wave per frame);
//   * no memory round trip on the recurrence: the utterance's log-probability rows are staged through LDS in chunks of
//     1024 floats (32 frames at V = 32), the next chunk's global loads in flight in registers while the current one is walked,
//     and a frame's emissions are read from LDS one frame ahead of their use;
//   * each thread keeps a CONTIGUOUS run of Q states in registers; the s-1 / s-2 neighbours of a run's ends come from the adjacent
//     lane as DPP wave shifts.  Up to 128 extended labels one wave does it all; beyond, FOUR waves (one per SIMD) split the states and
//     meet at one barrier per frame (ctc_ab_kernel's comment): 1.02 -> 0.56 us per frame at cfg5's 601 states;
//   * log2(2^a + 2^b + 2^c) as max3 / med3 / min3 + two v_exp_f32 + one v_log_f32.
// Numerics: ATen recurses on raw log-alpha, whose f32 values reach about -2000 at T = 624 (one ulp = 1.2e-4, so its own
// gradients carry ~1e-3 of noise).  Here both recursions run in the log2 domain on PER-FRAME-NORMALISED values: after every
// frame the values are shifted by the frame's maximum (the shifts are summed separately for the likelihood), so the stored
// numbers stay near 0 where f32 resolves ~1e-7, while tight alignments (T barely above S), whose end-state mass is 1e-40 of
// the prefix mass, neither underflow nor lose precision.  Posteriors are normalised per frame.
#include "a8_common.h"
#include "a8hip_internal.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
constexpr float NEG = -INFINITY;
constexpr int CTC_CHUNK = 1024;              // floats of log-probabilities staged per LDS buffer (V <= CTC_CHUNK: whole frames)

A8_DEV float lse2_3(float a, float b, float c) {           // log2(2^a + 2^b + 2^c), -inf safe
    const float m = fmaxf(a, fmaxf(b, c));                  // v_max3_f32
    const float lo = fminf(a, fminf(b, c));                 // v_min3_f32
    const float mid = __builtin_amdgcn_fmed3f(a, b, c);
    const float r = m + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(mid - m) + __builtin_amdgcn_exp2f(lo - m));
    return m == NEG ? NEG : r;                              // (m = -inf: the differences are NaN)
}
// whole-wave shifts by one lane as DPP modifiers (wave_shr:1 / wave_shl:1: gfx9 family only); the lane shifted in from outside reads 0 -- callers replace it
A8_DEV float wave_shr1(float v) { return dpp_f<0x138>(v); }
A8_DEV float wave_shl1(float v) { return dpp_f<0x130>(v); }
A8_DEV float lse2_2(float a, float b) {
    const float m = fmaxf(a, b), lo = fminf(a, b);
    const float r = m + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(lo - m));
    return m == NEG ? NEG : r;
}

// The staged log-probability rows of one utterance, walked forwards (alpha) or backwards (beta) by a block of NT threads.  V <= CTC_CHUNK:
// chunks of F = CTC_CHUNK / V whole frames (at most 64), double-buffered, chunk k + 1 requested (CTC_CHUNK / NT loads per thread, addresses
// clamped to the utterance's last row) before chunk k is walked and written to the other buffer behind it.  V > CTC_CHUNK (BPE
// vocabularies): one frame at a time, synchronously.
template <int NT> struct LpStage {
    static constexpr int R = CTC_CHUNK / NT;
    const float* lp; float* sm;
    long long rowstride;         // floats between frames of this utterance: B * V
    int V, F, Tmax, nchunk, big;
    float r[R];
    A8_DEV void init(const float* lp_b, float* sm_, int V_, int B, int Tmax_, int Tb) {
        lp = lp_b; sm = sm_; V = V_; Tmax = Tmax_; rowstride = (long long)B * V_;
        big = V_ > CTC_CHUNK;
        F = big ? 1 : min(64, CTC_CHUNK / V_);
        nchunk = (Tb + F - 1) / F;
    }
    A8_DEV void request(int k) {                                  // chunk k -> registers (small V only)
        const int tid = threadIdx.x, n = F * V;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int e = tid + NT * i;
            const int f = e / V, c = e - f * V;
            const int t = min(k * F + f, Tmax - 1);
            r[i] = e < n ? lp[(long long)t * rowstride + c] : 0.f;
        }
    }
    A8_DEV void commit(int buf) {                                 // registers -> LDS buffer, in log2 units
        const int tid = threadIdx.x, n = F * V;
        __syncthreads();                                          // every wave is done reading this buffer's previous contents
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int e = tid + NT * i;
            if (e < n) sm[buf * CTC_CHUNK + e] = r[i] * LOG2E;
        }
        __syncthreads();
    }
    A8_DEV const float* frame_big(int t) {                        // V > CTC_CHUNK: frame t synchronously into the single buffer
        __syncthreads();
        for (int c = threadIdx.x; c < V; c += NT) sm[c] = lp[(long long)t * rowstride + c] * LOG2E;
        __syncthreads();
        return sm;
    }
};

// dir = blockIdx.y: 0 = alpha (frames ascending), 1 = beta (descending).  A block of NW waves per (utterance, direction); thread tid keeps the
// Q CONSECUTIVE states tid Q .. tid Q + Q - 1 in registers.  ab / bb: [B][Tmax][pitch] normalised log2 values, state s at index s.
// status[b] = 1 when the utterance has a valid alignment.
// NW = 1 (2 S + 1 <= 128): the s - 1 / s - 2 neighbours of a run's ends come from the adjacent lane by wave shuffle; nothing else is shared.
// NW = 4 (longer label sequences; round 6, second pass): a frame of the one-wave kernel at 9-10 states per lane is ~300 instructions, 27 of them
// quarter-rate transcendentals, issued by ONE SIMD -- ~1 us per frame, 640 us at cfg5 with the other 255 CUs idle.  Four waves (one per SIMD of the CU)
// take a quarter of the states each and meet at ONE barrier per frame: before it every wave leaves its frame maximum and the two RAW values at its end
// that its neighbour needs (alpha: its last two states; beta: its first two) in a double-buffered LDS record; behind it every thread reads the four maxima
// (the frame's normalisation) and its neighbour wave's pair, which -- minus the maximum -- are its s - 1 / s - 2 (s + 1 / s + 2) inputs of the next frame.
template <int Q, int NW>
__global__ __launch_bounds__(64 * NW) void ctc_ab_kernel(const float* __restrict__ lp, const int32_t* __restrict__ targets,
                                                         const int32_t* __restrict__ tgt_off, const int32_t* __restrict__ in_len,
                                                         float* __restrict__ nll, float* __restrict__ abuf, float* __restrict__ bbuf,
                                                         int32_t* __restrict__ status, int Tmax, int B, int V, int pitch, int blank, int zero_infinity) {
    extern __shared__ float sm[];
    __shared__ float xch[2][NW][4];              // per frame parity and wave: {frame maximum of the wave, raw boundary value 1, 2}
    constexpr int NT = 64 * NW;
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = tgt_off[b + 1] - tgt_off[b];
    const int L = 2 * S + 1;
    const int Tb = min(in_len[b], Tmax);
    const int32_t* lab = targets + tgt_off[b];
    if (Tb <= 0 || L > NT * Q) {                 // (block-uniform)
        if (dir == 0 && tid == 0) { nll[b] = zero_infinity ? 0.f : INFINITY; status[b] = 0; }
        return;
    }
    float* const out = (dir == 0 ? abuf : bbuf) + (long long)b * Tmax * pitch;
    auto ext_of = [&](int s) { return (s < L && (s & 1)) ? lab[s >> 1] : blank; };

    int ext[Q];
    bool valid[Q], skip[Q];              // skip[j]: alpha may come from s - 2 (dir 0) / beta may go to s + 2 (dir 1)
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        const int s = tid * Q + j;
        valid[j] = s < L;
        ext[j] = ext_of(s);
        if (dir == 0) skip[j] = valid[j] && (s & 1) && s >= 3 && lab[s >> 1] != lab[(s >> 1) - 1];
        else skip[j] = (s + 2 < L) && (s & 1) && lab[(s + 2) >> 1] != lab[s >> 1];
    }
    // beta across waves: the emissions of the two states behind this thread's run (state (tid + 1) Q and the one after it; Q = 1: tid + 1, tid + 2)
    const int sq1 = (tid + 1) * Q, sq2 = Q >= 2 ? sq1 + 1 : tid + 2;
    const int extq1 = ext_of(sq1), extq2 = ext_of(sq2);
    LpStage<NT> st;
    st.init(lp + (long long)b * V, sm, V, B, Tmax, Tb);

    float a[Q];
    // alpha: the virtual frame -1 holds all mass in state 0 with the emission-free step into frame 0 (alpha_0(0) = p(blank),
    // alpha_0(1) = p(l_1)); beta_{T-1} (own emission excluded) = 1 on the last two states
    auto init_of = [&](int s) { return dir == 0 ? (s == 0 ? 0.f : NEG) : ((s < L && (s == L - 1 || s == L - 2)) ? 0.f : NEG); };
#pragma unroll
    for (int j = 0; j < Q; ++j) a[j] = init_of(tid * Q + j);
    // the neighbour WAVE's boundary pair, normalised: alpha -- the previous wave's last and second-last state; beta -- the next wave's first and second
    float x1 = NEG, x2 = NEG;
    if (NW > 1) {
        if (dir == 0 && w >= 1) { x1 = init_of(64 * Q * w - 1); x2 = init_of(64 * Q * w - 2); }
        if (dir == 1 && w + 1 < NW) { x1 = init_of(64 * Q * (w + 1)); x2 = init_of(64 * Q * (w + 1) + 1); }
    }
    int par = 0;
    float ll2 = 0.f;                         // log2-likelihood accumulated from the per-frame shifts (alpha)
    bool dead = false;

    // the frame's maximum over the block and (NW > 1) the neighbour wave's raw pair minus it; r1 / r2: this wave's raw pair (held by the lanes named below)
    auto frame_max = [&](float lm, float r1, float r2, bool first_pair) -> float {
        if constexpr (NW == 1) return lm;
        const int l1 = first_pair ? 0 : 63, l2 = Q >= 2 ? l1 : (first_pair ? 1 : 62);
        if (lane == l1) { xch[par][w][0] = lm; xch[par][w][1] = r1; }
        if (lane == l2) xch[par][w][2] = r2;
        __syncthreads();
        float mx = xch[par][0][0];
#pragma unroll
        for (int q = 1; q < NW; ++q) mx = fmaxf(mx, xch[par][q][0]);
        const int nbw = first_pair ? w + 1 : w - 1;          // beta reads the NEXT wave's first pair, alpha the PREVIOUS wave's last pair
        if (nbw >= 0 && nbw < NW) { x1 = xch[par][nbw][1] - mx; x2 = xch[par][nbw][2] - mx; }
        par ^= 1;
        return mx;
    };

    // The emissions of a frame: read UNCONDITIONALLY (ext[j] is the blank for the states beyond the utterance's last: always a valid index; a read
    // under `valid[j] ? .. : ..` gets its own basic block and its own s_waitcnt lgkmcnt(0)) and ONE FRAME AHEAD of their use wherever the next frame
    // sits in the same staged chunk -- an LDS round trip is as long as the rest of a frame's dependency chain.  em[Q], em[Q + 1]: beta's two states
    // behind the thread's run.
    // A state beyond the utterance's last gets the emission -inf: its alpha / beta then stay -inf through plain arithmetic, with no `valid ? .. : ..`
    // around the log-sum-exp (which the compiler turns into an exec-mask branch per state: ~10 scalar instructions and a taken-or-not branch each).
    auto emissions = [&](const float* pr, float (&em)[Q + 2]) {
#pragma unroll
        for (int j = 0; j < Q; ++j) { const float e = pr[ext[j]]; em[j] = valid[j] ? e : NEG; }
        if (dir == 1) {
            const float e1 = pr[extq1], e2 = pr[extq2];
            em[Q] = sq1 < L ? e1 : NEG; em[Q + 1] = sq2 < L ? e2 : NEG;
        }
    };
    auto alpha_frame = [&](const float (&em)[Q + 2], int t) {
        float na[Q];
        const float up1 = wave_shr1(a[Q - 1]);                    // lane l <- lane l - 1 (DPP: no LDS-crossbar round trip on the recurrence)
        const float up2 = Q >= 2 ? wave_shr1(a[Q >= 2 ? Q - 2 : 0]) : wave_shr1(up1);
        const float p1 = lane >= 1 ? up1 : x1;
        const float p2 = Q >= 2 ? (lane >= 1 ? up2 : x2) : (lane >= 2 ? up2 : (lane == 1 ? x1 : x2));
        float mx = NEG;
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            const float m1 = j >= 1 ? a[j >= 1 ? j - 1 : 0] : p1;
            const float m2 = j >= 2 ? a[j >= 2 ? j - 2 : 0] : (j == 1 ? p1 : p2);
            na[j] = lse2_3(a[j], m1, skip[j] ? m2 : NEG) + em[j];
            mx = fmaxf(mx, na[j]);
        }
        // raw pair for the next wave: this wave's last state (lane 63's na[Q-1]) and the one before it (Q >= 2: lane 63's na[Q-2]; Q = 1: lane 62's na[0])
        mx = frame_max(wave_max(mx), na[Q - 1], na[Q >= 2 ? Q - 2 : 0], false);
        if (mx == NEG) { dead = true; return; }                   // no alignment survives (block-uniform)
        ll2 += mx;
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            a[j] = na[j] - mx;
            if (valid[j]) out[(long long)t * pitch + tid * Q + j] = a[j];
        }
    };
    auto beta_frame = [&](const float (&em)[Q + 2], int t) {
#pragma unroll
        for (int j = 0; j < Q; ++j)
            if (valid[j]) out[(long long)t * pitch + tid * Q + j] = a[j];
        if (t == 0) return;
        // beta_{t-1}(s) = sum_{s' in {s, s+1, s+2 if skip}} beta_t(s') p_t(s')
        float wv[Q];
        const float eq1 = em[Q], eq2 = em[Q + 1];
#pragma unroll
        for (int j = 0; j < Q; ++j) wv[j] = a[j] + em[j];
        const float dn1 = wave_shl1(wv[0]);                       // lane l <- lane l + 1
        const float dn2 = Q >= 2 ? wave_shl1(wv[Q >= 2 ? 1 : 0]) : wave_shl1(dn1);
        // across the wave boundary: the next wave's first two states, (normalised beta_t) + their emission
        const float y1 = x1 + eq1;                                // state (tid + 1) Q     = the next wave's first state when lane == 63
        float q1, q2;
        if (Q >= 2) {
            const float y2 = x2 + eq2;
            q1 = lane < 63 ? dn1 : y1;
            q2 = lane < 63 ? dn2 : y2;
        } else {
            // Q = 1: state tid + 2 is the next wave's FIRST state for lane 62 and its SECOND for lane 63
            const float y2 = (lane == 62 ? x1 : x2) + eq2;
            q1 = lane < 63 ? dn1 : y1;
            q2 = lane < 62 ? dn2 : y2;
        }
        float nb[Q];
        float bm = NEG;
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            const float n1 = j + 1 < Q ? wv[j + 1 < Q ? j + 1 : 0] : q1;
            const float n2 = j + 2 < Q ? wv[j + 2 < Q ? j + 2 : 0] : (j + 1 < Q ? q1 : q2);
            nb[j] = lse2_3(wv[j], n1, skip[j] ? n2 : NEG);        // (a state beyond the last: every input is -inf)
            bm = fmaxf(bm, nb[j]);
        }
        // raw pair for the previous wave: this wave's first state (lane 0's nb[0]) and second (Q >= 2: lane 0's nb[1]; Q = 1: lane 1's nb[0])
        bm = frame_max(wave_max(bm), nb[0], nb[Q >= 2 ? 1 : 0], true);       // keep beta near 0 (any per-frame shift cancels in the posterior)
#pragma unroll
        for (int j = 0; j < Q; ++j) a[j] = bm == NEG ? NEG : nb[j] - bm;
        if (bm == NEG) { x1 = NEG; x2 = NEG; }
    };

    float em[Q + 2], en[Q + 2];
    if (st.big) {
        for (int i = 0; i < Tb && !dead; ++i) {
            const int t = dir == 0 ? i : Tb - 1 - i;
            emissions(st.frame_big(t), em);
            if (dir == 0) alpha_frame(em, t); else beta_frame(em, t);
        }
    } else {
        const int F = st.F, nch = st.nchunk;
        int k = dir == 0 ? 0 : nch - 1, buf = 0;
        st.request(k);
        st.commit(0);
        for (int c = 0; c < nch && !dead; ++c) {
            const int kn = dir == 0 ? k + 1 : k - 1;
            const bool more = c + 1 < nch;
            if (more) st.request(kn);                             // in flight while this chunk is walked
            const float* base = sm + buf * CTC_CHUNK;
            const int f_end = min(F, Tb - k * F);
            if (dir == 0) {
                emissions(base, em);
                for (int f = 0; f < f_end && !dead; ++f) {
                    if (f + 1 < f_end) emissions(base + (f + 1) * V, en);
                    alpha_frame(em, k * F + f);
#pragma unroll
                    for (int j = 0; j < Q + 2; ++j) em[j] = en[j];
                }
            } else {
                emissions(base + (f_end - 1) * V, em);
                for (int f = f_end - 1; f >= 0; --f) {
                    if (f >= 1) emissions(base + (f - 1) * V, en);
                    beta_frame(em, k * F + f);
#pragma unroll
                    for (int j = 0; j < Q + 2; ++j) em[j] = en[j];
                }
            }
            if (more) st.commit(buf ^ 1);
            buf ^= 1;
            k = kn;
        }
    }
    if (dir != 0) return;
    if (dead) {
        if (tid == 0) { nll[b] = zero_infinity ? 0.f : INFINITY; status[b] = 0; }
        return;
    }
    // termination: log2 Z = log2(alpha_{T-1}(L-1) + alpha_{T-1}(L-2)) in normalised units; the two states may sit in different waves
    float zt = NEG;
#pragma unroll
    for (int j = 0; j < Q; ++j) { const int s = tid * Q + j; if (valid[j] && (s == L - 1 || s == L - 2)) zt = lse2_2(zt, a[j]); }
    float zm = wave_max(zt);
    if constexpr (NW > 1) {
        __syncthreads();
        if (lane == 0) xch[0][w][0] = zm;
        __syncthreads();
        zm = xch[0][0][0];
#pragma unroll
        for (int q = 1; q < NW; ++q) zm = fmaxf(zm, xch[0][q][0]);
    }
    if (zm == NEG) {
        if (tid == 0) { nll[b] = zero_infinity ? 0.f : INFINITY; status[b] = 0; }
        return;
    }
    float zs = wave_sum(zt == NEG ? 0.f : exp2f(zt - zm));
    if constexpr (NW > 1) {
        if (lane == 0) xch[1][w][0] = zs;
        __syncthreads();
        zs = xch[1][0][0];
#pragma unroll
        for (int q = 1; q < NW; ++q) zs += xch[1][q][0];
    }
    if (tid == 0) { nll[b] = -(ll2 + zm + log2f(zs)) * LN2; status[b] = 1; }
}

// grad[t, b, :] = softmax probabilities of the frame - the posterior occupancy of its classes; one wave per (frame, utterance).
//   occupancy(c) = sum_{s: ext(s) = c} 2^(alpha_t(s) + beta_t(s)) / sum_s 2^(alpha_t(s) + beta_t(s))
// Frames beyond the input length and utterances without a valid alignment get zeros (zero_infinity).
__global__ __launch_bounds__(64) void ctc_grad_kernel(const float* __restrict__ lp, const int32_t* __restrict__ targets,
                                                      const int32_t* __restrict__ tgt_off, const int32_t* __restrict__ in_len,
                                                      const float* __restrict__ abuf, const float* __restrict__ bbuf,
                                                      const int32_t* __restrict__ status, float* __restrict__ grad,
                                                      int Tmax, int B, int V, int pitch, int blank) {
    extern __shared__ float occ[];           // [V]
    const int t = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    float* const grow = grad + ((long long)t * B + b) * V;
    const int Tb = min(in_len[b], Tmax);
    const int S = tgt_off[b + 1] - tgt_off[b], L = 2 * S + 1;
    if (t >= Tb || L > pitch || status[b] == 0) {
        for (int c = lane; c < V; c += 64) grow[c] = 0.f;
        return;
    }
    const float* ar = abuf + ((long long)b * Tmax + t) * pitch;
    const float* br = bbuf + ((long long)b * Tmax + t) * pitch;
    const int32_t* lab = targets + tgt_off[b];
    for (int c = lane; c < V; c += 64) occ[c] = 0.f;
    float mx = NEG;
    for (int s = lane; s < L; s += 64) mx = fmaxf(mx, ar[s] + br[s]);
    mx = wave_max(mx);                       // finite: the utterance has an alignment, so every frame carries mass
    __syncthreads();
    float sum = 0.f;
    for (int s = lane; s < L; s += 64) {
        const float g = ar[s] + br[s];
        if (g != NEG) {
            const float p = exp2f(g - mx);
            sum += p;
            atomicAdd(&occ[(s & 1) ? lab[s >> 1] : blank], p);
        }
    }
    const float inv = 1.0f / wave_sum(sum);
    __syncthreads();
    const float* prow = lp + ((long long)t * B + b) * V;
    for (int c = lane; c < V; c += 64) grow[c] = exp2f(prow[c] * LOG2E) - occ[c] * inv;
}

// ---- greedy decode (audio8/ctc.py:156-159: argmax over labels, unique_consecutive, drop blanks), one wave per
// utterance: lane = frame within a 64-frame chunk, neighbours by shuffle, survivors compacted with a ballot prefix.
template <typename T>
__global__ __launch_bounds__(64) void ctc_greedy_kernel(const T* __restrict__ lp, const int32_t* __restrict__ in_len,
                                                        int32_t* __restrict__ tokens, int32_t* __restrict__ out_len,
                                                        int Tn, int V, long long stride_b, long long stride_t, int blank) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = min(in_len[b], Tn);
    const T* base = lp + (long long)b * stride_b;
    int prev_last = -1, count = 0;
    for (int t0 = 0; t0 < n; t0 += 64) {
        const int t = t0 + lane;
        int best = -1;
        if (t < n) {
            const T* row = base + (long long)t * stride_t;
            float bv = to_f(row[0]);
            best = 0;
            for (int v = 1; v < V; ++v) {
                const float x = to_f(row[v]);
                if (x > bv) { bv = x; best = v; }          // first maximum wins, as torch.argmax
            }
        }
        int prev = __shfl_up(best, 1, 64);
        if (lane == 0) prev = prev_last;
        const bool keep = t < n && best != prev && best != blank;
        const unsigned long long m = __ballot(keep);
        if (keep) tokens[(long long)b * Tn + count + __popcll(m & ((1ull << lane) - 1ull))] = best;
        count += __popcll(m);
        prev_last = __shfl(best, 63, 64);
    }
    if (lane == 0) out_len[b] = count;
}

// ---- targets[B,S] padded with PAD / EOS -> the flat label list F.ctc_loss takes (audio8/ctc.py:193-194: masked_select of
// every entry that is neither PAD nor EOS, row-major) plus the per-utterance offsets (exclusive prefix sum of target_lengths).
// One block: wave w counts / compacts rows w, w + 4, ...; nothing goes through the host.
__global__ __launch_bounds__(256) void ctc_pack_kernel(const long long* __restrict__ targets, const int32_t* __restrict__ tlen,
                                                       int32_t* __restrict__ flat, int32_t* __restrict__ off, int32_t* __restrict__ status,
                                                       int B, int S, int pad, int eos) {
    extern __shared__ int cnt[];             // cnt[B]: label-list offset of each row (exclusive prefix of target_lengths)
    __shared__ int s_bad, s_max;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        int lrun = 0, mx = 0;
        for (int b = 0; b < B; ++b) { cnt[b] = lrun; off[b] = lrun; lrun += tlen[b]; mx = max(mx, tlen[b]); }
        off[B] = lrun;
        s_bad = 0; s_max = mx;
    }
    __syncthreads();
    // Every row lands at ITS offset and is cut at target_lengths[b]: a row whose PAD/EOS count disagrees with its stated length can
    // no longer shift the labels of the rows behind it (it is counted in status[0], which the host checks one call later).
    for (int b = w; b < B; b += 4) {
        int pos = 0;
        const int base = cnt[b], lim = tlen[b];
        for (int s0 = 0; s0 < S; s0 += 64) {
            const int s = s0 + lane;
            const long long v = s < S ? targets[(long long)b * S + s] : pad;
            const bool keep = s < S && v != pad && v != eos;
            const unsigned long long m = __ballot(keep);
            const int at = pos + __popcll(m & ((1ull << lane) - 1ull));
            if (keep && at < lim) flat[base + at] = (int32_t)v;
            pos += __popcll(m);
        }
        if (lane == 0 && pos != lim) atomicAdd(&s_bad, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0 && status) { status[0] = s_bad; status[1] = s_max; }
}

}  // namespace

extern "C" int a8_ctc_pack_targets(const int64_t* targets, const int32_t* target_lengths, int32_t* flat, int32_t* tgt_off, int32_t* status,
                                   int B, int S, int pad, int eos, void* stream) {
    A8_REQUIRE(B >= 1 && S >= 0 && B <= 8192, "a8_ctc_pack_targets: bad sizes B=%d S=%d", B, S);
    hipLaunchKernelGGL(ctc_pack_kernel, dim3(1), dim3(256), sizeof(int) * B, (hipStream_t)stream, (const long long*)targets, target_lengths,
                       flat, tgt_off, status, B, S, pad, eos);
    return a8_check_launch("a8_ctc_pack_targets");
}

extern "C" int a8_ctc_fwd_bwd(const float* lp, const int32_t* targets, const int32_t* tgt_off, const int32_t* in_len,
                              float* nll, float* grad, float* scratch, int T_, int B, int V, int Smax, int blank, int zero_infinity,
                              void* stream) {
    A8_REQUIRE(T_ >= 1 && B >= 1 && V >= 1, "a8_ctc_fwd_bwd: bad sizes T=%d B=%d V=%d", T_, B, V);
    A8_REQUIRE(blank >= 0 && blank < V, "a8_ctc_fwd_bwd: bad blank index");
    const int Lmax = 2 * Smax + 1;
    A8_REQUIRE(Smax >= 0 && Lmax <= 64 * 16, "a8_ctc_fwd_bwd: Smax=%d exceeds the supported 511 labels", Smax);
    A8_REQUIRE(V <= 8192, "a8_ctc_fwd_bwd: V=%d labels exceed the 8192 the per-frame LDS rows hold", V);
    A8_REQUIRE(B <= 65535, "a8_ctc_fwd_bwd: B=%d exceeds the grid's second dimension", B);
    const int spl = (Lmax + 63) / 64, pitch = 64 * spl;
    // scratch (A8_CTC_SCRATCH_FLOATS): alpha [B][T][pitch] | beta [B][T][pitch] | status i32 [B]; state s of a frame at index s
    const size_t plane = (size_t)B * T_ * pitch;
    float* abuf = scratch;
    float* bbuf = scratch + plane;
    int32_t* status = reinterpret_cast<int32_t*>(scratch + 2 * plane);
    const size_t shm_ab = sizeof(float) * (V > CTC_CHUNK ? (size_t)V : 2 * (size_t)CTC_CHUNK);
    hipStream_t s = (hipStream_t)stream;
    // one wave per (utterance, direction) up to 128 extended labels, four waves beyond (states per thread = ceil(Lmax / 256) <= 4)
#define A8_CTC_LAUNCH(Q, NW) hipLaunchKernelGGL((ctc_ab_kernel<Q, NW>), dim3(B, 2), dim3(64 * NW), shm_ab, s, lp, targets, tgt_off, in_len, nll, abuf, bbuf, \
                                                status, T_, B, V, pitch, blank, zero_infinity)
    if (Lmax <= 64) A8_CTC_LAUNCH(1, 1);
    else if (Lmax <= 128) A8_CTC_LAUNCH(2, 1);
    else if (Lmax <= 256) A8_CTC_LAUNCH(1, 4);
    else if (Lmax <= 512) A8_CTC_LAUNCH(2, 4);
    else if (Lmax <= 768) A8_CTC_LAUNCH(3, 4);
    else A8_CTC_LAUNCH(4, 4);
#undef A8_CTC_LAUNCH
    if (int e = a8_check_launch("a8_ctc_fwd_bwd (alpha / beta)")) return e;
    hipLaunchKernelGGL(ctc_grad_kernel, dim3(T_, B), dim3(64), sizeof(float) * V, s, lp, targets, tgt_off, in_len, abuf, bbuf, status, grad, T_, B, V, pitch, blank);
    return a8_check_launch("a8_ctc_fwd_bwd (gradient)");
}

extern "C" int a8_ctc_greedy_decode(int dtype, const void* lp, const int32_t* in_len, int32_t* tokens, int32_t* out_len,
                                    int B, int T_, int V, int64_t stride_b, int64_t stride_t, int blank, void* stream) {
    A8_REQUIRE(T_ >= 1 && B >= 1 && V >= 1, "a8_ctc_greedy_decode: bad sizes T=%d B=%d V=%d", T_, B, V);
    A8_REQUIRE(blank >= 0 && blank < V, "a8_ctc_greedy_decode: bad blank index");
    A8_REQUIRE(dtype == A8_F32 || dtype == A8_BF16, "a8_ctc_greedy_decode: bad dtype");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == A8_F32) hipLaunchKernelGGL(ctc_greedy_kernel<float>, dim3(B), dim3(64), 0, s, (const float*)lp, in_len, tokens, out_len, T_, V, (long long)stride_b, (long long)stride_t, blank);
    else hipLaunchKernelGGL(ctc_greedy_kernel<bf16>, dim3(B), dim3(64), 0, s, (const bf16*)lp, in_len, tokens, out_len, T_, V, (long long)stride_b, (long long)stride_t, blank);
    return a8_check_launch("a8_ctc_greedy_decode");
}
