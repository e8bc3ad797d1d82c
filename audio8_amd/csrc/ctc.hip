// a8hip: CTC loss forward + gradient (F.ctc_loss native path with cuDNN disabled, blank = Offsets.GO = 0,
// zero_infinity -- audio8/ctc.py:186-206).
//
// The alpha / beta recurrences over the 2S+1 extended labels are serial in time and tiny per step: the loss is LATENCY, and it
// sits between the model's forward and its backward with nothing to overlap it (round 5: one wave per utterance running alpha,
// then beta + gradient, with a global-memory round trip per frame in each: 6.5 ms of the 21.4 ms cfg5 step).  Round 6:
//   * alpha and beta run CONCURRENTLY, one wave each per utterance (grid B x 2), and only recurse; both store their per-frame
//     normalised log2 values; the posteriors and the gradient are a third, fully parallel launch (one wave per frame);
//   * no memory round trip on the recurrence: the utterance's log-probability rows are staged through LDS in chunks of
//     1024 floats (32 frames at V = 32), the next chunk's global loads in flight in registers while the current one is walked;
//   * each lane keeps a CONTIGUOUS run of SPL states in registers (SPL = ceil((2 Smax + 1) / 64), one instantiation per value),
//     the s-1 / s-2 neighbours of a run's ends come from the adjacent lane by wave shuffle;
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
A8_DEV float lse2_2(float a, float b) {
    const float m = fmaxf(a, b), lo = fminf(a, b);
    const float r = m + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(lo - m));
    return m == NEG ? NEG : r;
}

// The staged log-probability rows of one utterance, walked forwards (alpha) or backwards (beta).  V <= CTC_CHUNK: chunks of
// F = CTC_CHUNK / V whole frames (at most 64), double-buffered, chunk k + 1 requested (16 loads per lane, addresses clamped to the
// utterance's last row) before chunk k is walked and written to the other buffer behind it.  V > CTC_CHUNK (BPE vocabularies):
// one frame at a time, synchronously.
struct LpStage {
    const float* lp; float* sm;
    long long rowstride;         // floats between frames of this utterance: B * V
    int V, F, Tmax, nchunk, big;
    float r[16];
    A8_DEV void init(const float* lp_b, float* sm_, int V_, int B, int Tmax_, int Tb) {
        lp = lp_b; sm = sm_; V = V_; Tmax = Tmax_; rowstride = (long long)B * V_;
        big = V_ > CTC_CHUNK;
        F = big ? 1 : min(64, CTC_CHUNK / V_);
        nchunk = (Tb + F - 1) / F;
    }
    A8_DEV void request(int k) {                                  // chunk k -> registers (small V only)
        const int lane = threadIdx.x, n = F * V;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int e = lane + 64 * i;
            const int f = e / V, c = e - f * V;
            const int t = min(k * F + f, Tmax - 1);
            r[i] = e < n ? lp[(long long)t * rowstride + c] : 0.f;
        }
    }
    A8_DEV void commit(int buf) {                                 // registers -> LDS buffer, in log2 units
        const int lane = threadIdx.x, n = F * V;
        __syncthreads();                                          // (one wave: orders the LDS reads of this buffer's last user)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int e = lane + 64 * i;
            if (e < n) sm[buf * CTC_CHUNK + e] = r[i] * LOG2E;
        }
        __syncthreads();
    }
    A8_DEV const float* frame_big(int t) {                        // V > CTC_CHUNK: frame t synchronously into the single buffer
        __syncthreads();
        for (int c = threadIdx.x; c < V; c += 64) sm[c] = lp[(long long)t * rowstride + c] * LOG2E;
        __syncthreads();
        return sm;
    }
};

// dir = blockIdx.y: 0 = alpha (frames ascending), 1 = beta (descending).  ab / bb: [B][Tmax][64 SPL] normalised log2 values, element
// (lane, j) = state lane * SPL + j at index j * 64 + lane (coalesced).  status[b] = 1 when the utterance has a valid alignment.
template <int SPL>
__global__ __launch_bounds__(64) void ctc_ab_kernel(const float* __restrict__ lp, const int32_t* __restrict__ targets,
                                                    const int32_t* __restrict__ tgt_off, const int32_t* __restrict__ in_len,
                                                    float* __restrict__ nll, float* __restrict__ abuf, float* __restrict__ bbuf,
                                                    int32_t* __restrict__ status, int Tmax, int B, int V, int blank, int zero_infinity) {
    extern __shared__ float sm[];
    const int b = blockIdx.x, dir = blockIdx.y, lane = threadIdx.x;
    const int S = tgt_off[b + 1] - tgt_off[b];
    const int L = 2 * S + 1;
    const int Tb = min(in_len[b], Tmax);
    const int32_t* lab = targets + tgt_off[b];
    if (Tb <= 0 || L > 64 * SPL) {
        if (dir == 0 && lane == 0) { nll[b] = zero_infinity ? 0.f : INFINITY; status[b] = 0; }
        return;
    }
    constexpr int PITCH = 64 * SPL;
    float* const out = (dir == 0 ? abuf : bbuf) + (long long)b * Tmax * PITCH;

    int ext[SPL];
    bool valid[SPL], skip[SPL];          // skip[j]: alpha may come from s - 2 (dir 0) / beta may go to s + 2 (dir 1)
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        const int s = lane * SPL + j;
        valid[j] = s < L;
        ext[j] = (valid[j] && (s & 1)) ? lab[s >> 1] : blank;
        if (dir == 0) skip[j] = valid[j] && (s & 1) && s >= 3 && lab[s >> 1] != lab[(s >> 1) - 1];
        else skip[j] = (s + 2 < L) && (s & 1) && lab[(s + 2) >> 1] != lab[s >> 1];
    }
    LpStage st;
    st.init(lp + (long long)b * V, sm, V, B, Tmax, Tb);

    float a[SPL];
    // alpha: the virtual frame -1 holds all mass in state 0 with the emission-free step into frame 0 (alpha_0(0) = p(blank),
    // alpha_0(1) = p(l_1)); beta_{T-1} (own emission excluded) = 1 on the last two states
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        const int s = lane * SPL + j;
        a[j] = dir == 0 ? (s == 0 ? 0.f : NEG) : ((valid[j] && (s == L - 1 || s == L - 2)) ? 0.f : NEG);
    }
    float ll2 = 0.f;                         // log2-likelihood accumulated from the per-frame shifts (alpha)
    bool dead = false;

    auto alpha_frame = [&](const float* pr, int t) {
        const float up1 = __shfl_up(a[SPL - 1], 1, 64);
        const float up2 = SPL >= 2 ? __shfl_up(a[SPL >= 2 ? SPL - 2 : 0], 1, 64) : __shfl_up(a[0], 2, 64);
        const float p1 = lane >= 1 ? up1 : NEG;
        const float p2 = SPL >= 2 ? (lane >= 1 ? up2 : NEG) : (lane >= 2 ? up2 : NEG);
        float na[SPL];
        float mx = NEG;
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const float m1 = j >= 1 ? a[j >= 1 ? j - 1 : 0] : p1;
            const float m2 = j >= 2 ? a[j >= 2 ? j - 2 : 0] : (j == 1 ? p1 : p2);
            const float v = lse2_3(a[j], m1, skip[j] ? m2 : NEG);
            na[j] = valid[j] ? v + pr[ext[j]] : NEG;
            mx = fmaxf(mx, na[j]);
        }
        mx = wave_max(mx);
        if (mx == NEG) { dead = true; return; }                   // no alignment survives (wave-uniform)
        ll2 += mx;
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            a[j] = na[j] - mx;
            out[(long long)t * PITCH + j * 64 + lane] = a[j];
        }
    };
    auto beta_frame = [&](const float* pr, int t) {
#pragma unroll
        for (int j = 0; j < SPL; ++j) out[(long long)t * PITCH + j * 64 + lane] = a[j];
        if (t == 0) return;
        // beta_{t-1}(s) = sum_{s' in {s, s+1, s+2 if skip}} beta_t(s') p_t(s')
        float w[SPL];
#pragma unroll
        for (int j = 0; j < SPL; ++j) w[j] = valid[j] ? a[j] + pr[ext[j]] : NEG;
        const float dn1 = __shfl_down(w[0], 1, 64);
        const float dn2 = SPL >= 2 ? __shfl_down(w[SPL >= 2 ? 1 : 0], 1, 64) : __shfl_down(w[0], 2, 64);
        const float q1 = lane < 63 ? dn1 : NEG;
        const float q2 = SPL >= 2 ? (lane < 63 ? dn2 : NEG) : (lane < 62 ? dn2 : NEG);
        float nb[SPL];
        float bm = NEG;
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const float n1 = j + 1 < SPL ? w[j + 1 < SPL ? j + 1 : 0] : q1;
            const float n2 = j + 2 < SPL ? w[j + 2 < SPL ? j + 2 : 0] : (j + 1 < SPL ? q1 : q2);
            nb[j] = valid[j] ? lse2_3(w[j], n1, skip[j] ? n2 : NEG) : NEG;
            bm = fmaxf(bm, nb[j]);
        }
        bm = wave_max(bm);                     // keep beta near 0 (any per-frame shift cancels in the posterior)
#pragma unroll
        for (int j = 0; j < SPL; ++j) a[j] = bm == NEG ? NEG : nb[j] - bm;
    };

    if (st.big) {
        for (int i = 0; i < Tb && !dead; ++i) {
            const int t = dir == 0 ? i : Tb - 1 - i;
            const float* pr = st.frame_big(t);
            if (dir == 0) alpha_frame(pr, t); else beta_frame(pr, t);
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
                for (int f = 0; f < f_end && !dead; ++f) alpha_frame(base + f * V, k * F + f);
            } else {
                for (int f = f_end - 1; f >= 0; --f) beta_frame(base + f * V, k * F + f);
            }
            if (more) st.commit(buf ^ 1);
            buf ^= 1;
            k = kn;
        }
    }
    if (dir != 0) return;
    if (dead) {
        if (lane == 0) { nll[b] = zero_infinity ? 0.f : INFINITY; status[b] = 0; }
        return;
    }
    // termination: log2 Z = log2(alpha_{T-1}(L-1) + alpha_{T-1}(L-2)) in normalised units
    float zt = NEG;
#pragma unroll
    for (int j = 0; j < SPL; ++j) { const int s = lane * SPL + j; if (valid[j] && (s == L - 1 || s == L - 2)) zt = lse2_2(zt, a[j]); }
    const float zm = wave_max(zt);
    if (zm == NEG) {
        if (lane == 0) { nll[b] = zero_infinity ? 0.f : INFINITY; status[b] = 0; }
        return;
    }
    const float z2 = zm + log2f(wave_sum(zt == NEG ? 0.f : exp2f(zt - zm)));
    if (lane == 0) { nll[b] = -(ll2 + z2) * LN2; status[b] = 1; }
}

// grad[t, b, :] = softmax probabilities of the frame - the posterior occupancy of its classes; one wave per (frame, utterance).
//   occupancy(c) = sum_{s: ext(s) = c} 2^(alpha_t(s) + beta_t(s)) / sum_s 2^(alpha_t(s) + beta_t(s))
// Frames beyond the input length and utterances without a valid alignment get zeros (zero_infinity).
__global__ __launch_bounds__(64) void ctc_grad_kernel(const float* __restrict__ lp, const int32_t* __restrict__ targets,
                                                      const int32_t* __restrict__ tgt_off, const int32_t* __restrict__ in_len,
                                                      const float* __restrict__ abuf, const float* __restrict__ bbuf,
                                                      const int32_t* __restrict__ status, float* __restrict__ grad,
                                                      int Tmax, int B, int V, int spl, int blank) {
    extern __shared__ float occ[];           // [V]
    const int t = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    float* const grow = grad + ((long long)t * B + b) * V;
    const int Tb = min(in_len[b], Tmax);
    const int S = tgt_off[b + 1] - tgt_off[b], L = 2 * S + 1;
    if (t >= Tb || L > 64 * spl || status[b] == 0) {
        for (int c = lane; c < V; c += 64) grow[c] = 0.f;
        return;
    }
    const int pitch = 64 * spl;
    const float* ar = abuf + ((long long)b * Tmax + t) * pitch;
    const float* br = bbuf + ((long long)b * Tmax + t) * pitch;
    const int32_t* lab = targets + tgt_off[b];
    for (int c = lane; c < V; c += 64) occ[c] = 0.f;
    float mx = NEG;
    for (int j = 0; j < spl; ++j) {
        const int s = lane * spl + j;
        if (s < L) mx = fmaxf(mx, ar[j * 64 + lane] + br[j * 64 + lane]);
    }
    mx = wave_max(mx);                       // finite: the utterance has an alignment, so every frame carries mass
    __syncthreads();
    float sum = 0.f;
    for (int j = 0; j < spl; ++j) {
        const int s = lane * spl + j;
        if (s < L) {
            const float g = ar[j * 64 + lane] + br[j * 64 + lane];
            if (g != NEG) {
                const float p = exp2f(g - mx);
                sum += p;
                atomicAdd(&occ[(s & 1) ? lab[s >> 1] : blank], p);
            }
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
    const int spl = (Lmax + 63) / 64;
    // scratch (A8_CTC_SCRATCH_FLOATS): alpha [B][T][64 spl] | beta [B][T][64 spl] | status i32 [B]
    const size_t plane = (size_t)B * T_ * 64 * spl;
    float* abuf = scratch;
    float* bbuf = scratch + plane;
    int32_t* status = reinterpret_cast<int32_t*>(scratch + 2 * plane);
    const size_t shm_ab = sizeof(float) * (V > CTC_CHUNK ? (size_t)V : 2 * (size_t)CTC_CHUNK);
    hipStream_t s = (hipStream_t)stream;
#define A8_CTC_LAUNCH(N) case N: hipLaunchKernelGGL(ctc_ab_kernel<N>, dim3(B, 2), dim3(64), shm_ab, s, lp, targets, tgt_off, in_len, nll, abuf, bbuf, status, T_, B, V, blank, zero_infinity); break;
    switch (spl) {
        A8_CTC_LAUNCH(1) A8_CTC_LAUNCH(2) A8_CTC_LAUNCH(3) A8_CTC_LAUNCH(4) A8_CTC_LAUNCH(5) A8_CTC_LAUNCH(6) A8_CTC_LAUNCH(7) A8_CTC_LAUNCH(8)
        A8_CTC_LAUNCH(9) A8_CTC_LAUNCH(10) A8_CTC_LAUNCH(11) A8_CTC_LAUNCH(12) A8_CTC_LAUNCH(13) A8_CTC_LAUNCH(14) A8_CTC_LAUNCH(15)
        default: hipLaunchKernelGGL(ctc_ab_kernel<16>, dim3(B, 2), dim3(64), shm_ab, s, lp, targets, tgt_off, in_len, nll, abuf, bbuf, status, T_, B, V, blank, zero_infinity); break;
    }
#undef A8_CTC_LAUNCH
    if (int e = a8_check_launch("a8_ctc_fwd_bwd (alpha / beta)")) return e;
    hipLaunchKernelGGL(ctc_grad_kernel, dim3(T_, B), dim3(64), sizeof(float) * V, s, lp, targets, tgt_off, in_len, abuf, bbuf, status, grad, T_, B, V, spl, blank);
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
