// a8hip: CTC loss forward + gradient, one wavefront per utterance (F.ctc_loss native path with cuDNN
// disabled, blank = Offsets.GO = 0, zero_infinity -- audio8/ctc.py:186-206).
//
// The alpha/beta recurrences over the 2S+1 extended labels are serial in time and tiny per step, so the
// kernel is latency-bound: each lane keeps a CONTIGUOUS run of SPL states in registers, the s-1 / s-2
// neighbours of a run's first states come from the previous lane by wave shuffle, and nothing but the
// per-frame class log-probabilities passes through LDS.
//
// Numerics: ATen recurses on raw log-alpha, whose f32 values reach about -2000 at T = 624 (one ulp = 1.2e-4, so
// its own gradients carry ~1e-3 of noise).  Here the recursion runs in the log2 domain on PER-FRAME-NORMALISED
// values: after every frame log2-alpha is shifted by the frame's log-sum (kept separately), so the stored numbers
// stay near 0 where f32 resolves ~1e-7, while tight alignments (T barely above S), whose end-state mass is
// 1e-40 of the prefix mass, neither underflow nor lose precision.  Posteriors are normalised per frame.
#include "a8_common.h"
#include "a8hip_internal.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
constexpr float NEG = -INFINITY;

A8_DEV float lse2_3(float a, float b, float c) {           // log2(2^a + 2^b + 2^c), -inf safe
    const float m = fmaxf(a, fmaxf(b, c));
    if (m == NEG) return NEG;
    return m + log2f(exp2f(a - m) + exp2f(b - m) + exp2f(c - m));
}
A8_DEV float wave_lse2(float v) {                            // log2 sum_lanes 2^v, -inf safe
    const float m = wave_max(v);
    if (m == NEG) return NEG;
    return m + log2f(wave_sum(exp2f(v - m)));
}

template <int SPL>
__global__ __launch_bounds__(64) void ctc_kernel(const float* __restrict__ lp, const int32_t* __restrict__ targets,
                                                 const int32_t* __restrict__ tgt_off, const int32_t* __restrict__ in_len,
                                                 float* __restrict__ nll, float* __restrict__ grad, float* __restrict__ abuf,
                                                 int Tmax, int B, int V, int Lmax, int blank, int zero_infinity) {
    extern __shared__ float sm[];            // pr[V] (log2 class probabilities of the frame) | occ[V]
    float* pr = sm;
    float* occ = sm + V;
    const int b = blockIdx.x, lane = threadIdx.x;
    const int S = tgt_off[b + 1] - tgt_off[b];
    const int L = 2 * S + 1;
    int Tb = in_len[b];
    if (Tb > Tmax) Tb = Tmax;
    const int32_t* lab = targets + tgt_off[b];
    float* ab = abuf + (long long)b * Tmax * Lmax;       // normalised log2-alpha of every frame

    // gradient defaults to zero (frames beyond the input length, infeasible utterances)
    for (int t = 0; t < Tmax; ++t)
        for (int c = lane; c < V; c += 64) grad[((long long)t * B + b) * V + c] = 0.f;
    if (Tb <= 0 || L > 64 * SPL) { if (lane == 0) nll[b] = zero_infinity ? 0.f : INFINITY; return; }

    int ext[SPL];
    bool skip[SPL];
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        const int s = lane * SPL + j;
        ext[j] = (s < L && (s & 1)) ? lab[s >> 1] : blank;
        skip[j] = (s < L) && (s & 1) && s >= 3 && lab[s >> 1] != lab[(s >> 1) - 1];
    }
    float a[SPL];
#pragma unroll
    for (int j = 0; j < SPL; ++j) a[j] = NEG;
    float ll2 = 0.f;                         // log2-likelihood accumulated from the per-frame normalisers
    // ---------------- forward
    for (int t = 0; t < Tb; ++t) {
        __syncthreads();
        for (int c = lane; c < V; c += 64) pr[c] = lp[((long long)t * B + b) * V + c] * LOG2E;
        __syncthreads();
        float na[SPL];
        if (t == 0) {
#pragma unroll
            for (int j = 0; j < SPL; ++j) { const int s = lane * SPL + j; na[j] = (s < 2 && s < L) ? pr[ext[j]] : NEG; }
        } else {
            const float up1 = __shfl_up(a[SPL - 1], 1, 64);
            const float up2 = SPL >= 2 ? __shfl_up(a[SPL >= 2 ? SPL - 2 : 0], 1, 64) : __shfl_up(a[0], 2, 64);
            const float p1 = lane >= 1 ? up1 : NEG;
            const float p2 = SPL >= 2 ? (lane >= 1 ? up2 : NEG) : (lane >= 2 ? up2 : NEG);
#pragma unroll
            for (int j = 0; j < SPL; ++j) {
                const int s = lane * SPL + j;
                const float m1 = j >= 1 ? a[j >= 1 ? j - 1 : 0] : p1;
                const float m2 = j >= 2 ? a[j >= 2 ? j - 2 : 0] : (j == 1 ? p1 : p2);
                const float v = lse2_3(a[j], m1, skip[j] ? m2 : NEG);
                na[j] = s < L ? v + pr[ext[j]] : NEG;
            }
        }
        float mx = NEG;
#pragma unroll
        for (int j = 0; j < SPL; ++j) mx = fmaxf(mx, na[j]);
        mx = wave_max(mx);
        float sum = 0.f;
        if (mx != NEG) {
#pragma unroll
            for (int j = 0; j < SPL; ++j) sum += exp2f(na[j] - mx);
        }
        sum = wave_sum(sum);
        const float c2 = mx == NEG ? NEG : mx + log2f(sum);          // log2 of the frame's total mass
        if (c2 == NEG) {                                             // no alignment survives: infeasible
            if (lane == 0) nll[b] = zero_infinity ? 0.f : INFINITY;
            return;
        }
        ll2 += c2;
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            a[j] = na[j] - c2;
            const int s = lane * SPL + j;
            if (s < L) ab[(long long)t * Lmax + s] = a[j];
        }
    }
    // termination: log2 Z = log2(alpha_{T-1}(L-1) + alpha_{T-1}(L-2)) in normalised units
    float zt = NEG;
#pragma unroll
    for (int j = 0; j < SPL; ++j) { const int s = lane * SPL + j; if (s == L - 1 || s == L - 2) zt = zt == NEG ? a[j] : lse2_3(zt, a[j], NEG); }
    const float z2 = wave_lse2(zt);
    if (z2 == NEG) {
        if (lane == 0) nll[b] = zero_infinity ? 0.f : INFINITY;
        return;
    }
    if (lane == 0) nll[b] = -(ll2 + z2) * LN2;
    // ---------------- backward + gradient (beta excludes the emission of its own frame; re-normalised per frame)
    float bt[SPL];
#pragma unroll
    for (int j = 0; j < SPL; ++j) { const int s = lane * SPL + j; bt[j] = (s < L && (s == L - 1 || s == L - 2)) ? 0.f : NEG; }
    for (int t = Tb - 1; t >= 0; --t) {
        __syncthreads();
        for (int c = lane; c < V; c += 64) { pr[c] = lp[((long long)t * B + b) * V + c] * LOG2E; occ[c] = 0.f; }
        __syncthreads();
        // posterior occupancy of frame t: 2^(alpha + beta) normalised over the states
        float g2[SPL];
        float mx = NEG;
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const int s = lane * SPL + j;
            g2[j] = s < L ? ab[(long long)t * Lmax + s] + bt[j] : NEG;
            mx = fmaxf(mx, g2[j]);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < SPL; ++j) { g2[j] = g2[j] == NEG ? 0.f : exp2f(g2[j] - mx); sum += g2[j]; }
        const float inv = 1.0f / wave_sum(sum);
#pragma unroll
        for (int j = 0; j < SPL; ++j)
            if (g2[j] != 0.f) atomicAdd(&occ[ext[j]], g2[j] * inv);
        __syncthreads();
        for (int c = lane; c < V; c += 64) grad[((long long)t * B + b) * V + c] = exp2f(pr[c]) - occ[c];
        if (t == 0) break;
        // beta_{t-1}(s) = sum_{s' in {s, s+1, s+2 if skip(s+2)}} beta_t(s') p_t(s'), then shift by the frame's log-sum
        float w[SPL];
        bool sk[SPL];
#pragma unroll
        for (int j = 0; j < SPL; ++j) { w[j] = bt[j] + pr[ext[j]]; sk[j] = skip[j]; }
        const float dn1 = __shfl_down(w[0], 1, 64);
        const float dn2 = SPL >= 2 ? __shfl_down(w[SPL >= 2 ? 1 : 0], 1, 64) : __shfl_down(w[0], 2, 64);
        const int dsk1i = __shfl_down((int)sk[0], 1, 64);
        const int dsk2i = SPL >= 2 ? __shfl_down((int)sk[SPL >= 2 ? 1 : 0], 1, 64) : __shfl_down((int)sk[0], 2, 64);
        const bool has1 = lane < 63, has2 = SPL >= 2 ? lane < 63 : lane < 62;
        float nb[SPL];
        float bm = NEG;
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const int s = lane * SPL + j;
            const float n1 = j + 1 < SPL ? w[j + 1 < SPL ? j + 1 : 0] : (has1 ? dn1 : NEG);
            float n2; bool s2;
            if (j + 2 < SPL) { n2 = w[j + 2 < SPL ? j + 2 : 0]; s2 = sk[j + 2 < SPL ? j + 2 : 0]; }
            else if (j + 1 < SPL) { n2 = has1 ? dn1 : NEG; s2 = has1 && dsk1i; }       // j == SPL-2: next lane's state 0
            else { n2 = has2 ? dn2 : NEG; s2 = has2 && dsk2i; }                          // next lane's state 1 (or lane+2's, SPL == 1)
            nb[j] = s < L ? lse2_3(w[j], n1, s2 ? n2 : NEG) : NEG;
            bm = fmaxf(bm, nb[j]);
        }
        bm = wave_max(bm);                     // keep beta near 0: subtract the frame maximum (any shift cancels in the posterior)
#pragma unroll
        for (int j = 0; j < SPL; ++j) bt[j] = nb[j] == NEG ? NEG : nb[j] - bm;
    }
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
                              float* nll, float* grad, float* alpha, int T_, int B, int V, int Smax, int blank, int zero_infinity,
                              void* stream) {
    A8_REQUIRE(T_ >= 1 && B >= 1 && V >= 1, "a8_ctc_fwd_bwd: bad sizes T=%d B=%d V=%d", T_, B, V);
    A8_REQUIRE(blank >= 0 && blank < V, "a8_ctc_fwd_bwd: bad blank index");
    const int Lmax = 2 * Smax + 1;
    A8_REQUIRE(Smax >= 0 && Lmax <= 64 * 16, "a8_ctc_fwd_bwd: Smax=%d exceeds the supported 511 labels", Smax);
    const int spl = (Lmax + 63) / 64;
    const size_t shm = sizeof(float) * 2 * V;
    A8_REQUIRE(shm <= 64 * 1024, "a8_ctc_fwd_bwd: V=%d labels exceed the 8192 the per-frame LDS rows hold", V);
    hipStream_t s = (hipStream_t)stream;
#define A8_CTC_LAUNCH(N) hipLaunchKernelGGL(ctc_kernel<N>, dim3(B), dim3(64), shm, s, lp, targets, tgt_off, in_len, nll, grad, alpha, T_, B, V, Lmax, blank, zero_infinity)
    if (spl <= 1) A8_CTC_LAUNCH(1);
    else if (spl <= 2) A8_CTC_LAUNCH(2);
    else if (spl <= 4) A8_CTC_LAUNCH(4);
    else if (spl <= 8) A8_CTC_LAUNCH(8);
    else A8_CTC_LAUNCH(16);
#undef A8_CTC_LAUNCH
    return a8_check_launch("a8_ctc_fwd_bwd");
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
