// a8hip: fixed launch sequences of the path behind ONE C call each (host code only: every kernel is launched through the
// library's own entry points, in the order audio8_amd/ops.py used to issue them one ctypes call at a time).  The point is
// the host: twelve Python -> C transitions per encoder layer forward become one, so that the launching thread stays well ahead of
// the GPU when eight ranks share a node.
#include <cmath>
#include <cstring>

#include "a8_common.h"
#include "a8hip_internal.h"

namespace {

a8_gemm_desc nt_desc(int dtype, const void* A, const void* B, void* C, int M, int N, int K, const float* bias) {
    a8_gemm_desc d;
    std::memset(&d, 0, sizeof(d));
    d.size = (int)sizeof(d);
    d.dtype = dtype; d.layout = A8_GEMM_NT;
    d.M = M; d.N = N; d.K = K;
    d.A = A; d.lda = K;
    d.B = B; d.ldb = K;
    d.C = C; d.ldc = N;
    d.bias = bias;
    d.ldz = N;
    d.splitk = 1;
    d.alpha = 1.0f;
    return d;
}

}  // namespace

// One eight_mile TransformerEncoder forward, post-LN (audio8/wav2vec2.py:613-622):
//   qkv = x Wqkv^T + b;  o = attention(q, k, v);  ao = o Wo^T + bo;  x1 = LN2(x + drop(ao));
//   h = gelu(z1 = x1 W1^T + b1);  f = h W2^T + b2;  x2 = LN1(x1 + drop(f))
extern "C" int a8_encoder_layer_fwd(const a8_encoder_layer_desc* e, void* stream) {
    A8_REQUIRE(e != nullptr, "a8_encoder_layer_fwd: null descriptor");
    A8_REQUIRE(e->size == (int)sizeof(*e), "a8_encoder_layer_fwd: descriptor size %d, expected %d", e->size, (int)sizeof(*e));
    A8_REQUIRE(e->B > 0 && e->T > 0 && e->H > 0 && e->d == e->H * 64 && e->dff > 0 && e->b0 >= 0, "a8_encoder_layer_fwd: bad sizes B=%d T=%d H=%d d=%d dff=%d b0=%d",
               e->B, e->T, e->H, e->d, e->dff, e->b0);
    const int M = e->B * e->T, d = e->d, esz = e->dtype == A8_BF16 ? 2 : 4;
    a8_gemm_desc g = nt_desc(e->dtype, e->x, e->wqkv, e->qkv, M, 3 * d, d, e->bqkv);
    if (int rc = a8_gemm(&g, stream)) return rc;
    const char* q = (const char*)e->qkv;
    // e->b0: this call covers utterances [b0, b0 + B) of a larger logical batch (the two halves of a batch on two streams): the
    // dropout counters start there, so forward draws the masks the whole-batch backward regenerates
    if (int rc = a8i_attn_fwd(e->dtype, q, q + (size_t)d * esz, q + (size_t)2 * d * esz, e->o, e->lse, e->keymask, e->B, e->T, e->H, 3 * d, d,
                              1.0f / std::sqrt((float)(d / e->H)), e->p, e->seed_attn, e->b0 * e->H, stream))
        return rc;
    g = nt_desc(e->dtype, e->o, e->wo, e->ao, M, d, d, e->bo);
    if (int rc = a8_gemm(&g, stream)) return rc;
    const int64_t rng0 = (int64_t)e->b0 * e->T * d;
    if (int rc = a8i_layernorm_fwd(e->dtype, e->ao, e->x, e->g2, e->be2, e->x1, e->mean2, e->rstd2, M, d, e->eps, e->p, e->seed1, 3, rng0, stream)) return rc;
    g = nt_desc(e->dtype, e->x1, e->w1, e->h, M, e->dff, d, e->b1);
    g.act = A8_ACT_GELU; g.zout = e->z1;
    if (int rc = a8_gemm(&g, stream)) return rc;
    g = nt_desc(e->dtype, e->h, e->w2, e->f, M, d, e->dff, e->b2);
    if (int rc = a8_gemm(&g, stream)) return rc;
    return a8i_layernorm_fwd(e->dtype, e->f, e->x1, e->g1, e->be1, e->x2, e->mean1, e->rstd1, M, d, e->eps, e->p, e->seed2, 4, rng0, stream);
}

// The data-gradient chain of the same layer's backward up to and including the attention core, on one stream (the weight gradients
// and the final QKV data gradient stay with the caller: they go to a side stream / follow the side stream's hand-over):
//   LN1 backward (+ b2 gradient) -> dz1 = (df W2) * gelu'(z1) (+ b1 gradient) -> dx1_ffn = dz1 W1 -> LN2 backward (+ bo gradient)
//   -> do = dao Wo -> attention backward (dq | dk | dv into dqkv).
// With p = 0 the dropout-free LayerNorms hand back ONE gradient for both of their inputs: dx1_res aliases df, dx_res aliases dao
// (pass the same pointers).
extern "C" int a8_encoder_layer_bwd(const a8_encoder_layer_bwd_desc* e, void* stream) {
    A8_REQUIRE(e != nullptr, "a8_encoder_layer_bwd: null descriptor");
    A8_REQUIRE(e->size == (int)sizeof(*e), "a8_encoder_layer_bwd: descriptor size %d, expected %d", e->size, (int)sizeof(*e));
    A8_REQUIRE(e->B > 0 && e->T > 0 && e->H > 0 && e->d == e->H * 64 && e->dff > 0, "a8_encoder_layer_bwd: bad sizes B=%d T=%d H=%d d=%d dff=%d",
               e->B, e->T, e->H, e->d, e->dff);
    A8_REQUIRE(e->p > 0.f || (e->dx1_res == e->df && e->dx_res == e->dao), "a8_encoder_layer_bwd: with p = 0, dx1_res / dx_res must alias df / dao");
    A8_REQUIRE((e->ln1_partials == nullptr) == (e->ln2_partials == nullptr), "a8_encoder_layer_bwd: ln1_partials and ln2_partials go together");
    const int M = e->B * e->T, d = e->d, esz = e->dtype == A8_BF16 ? 2 : 4;
    const bool drop = e->p > 0.f;
    const int64_t rng0 = (int64_t)e->b0 * e->T * d;          // dropout counters of utterances [b0, b0 + B) of the logical batch (see the forward)
    // e->ln1_partials / ln2_partials (both or neither): the LayerNorm parameter gradients and the w_2 / w_O bias gradients leave the
    // kernels as per-block partial rows (a8_colsum_partials_reduce, issued by the caller off this stream, adds them into dg / dbe / db)
    if (int rc = a8i_layernorm_bwd(e->dtype, e->dx2, nullptr, e->f, e->x1, e->g1, e->mean1, e->rstd1, e->df, drop ? e->dx1_res : nullptr,
                                   e->dg1, e->dbe1, e->db2, M, d, e->p, e->seed2, 4, rng0, stream, e->ln1_partials))
        return rc;
    a8_gemm_desc g = nt_desc(e->dtype, e->df, e->w2_t, e->dz1, M, e->dff, d, nullptr);
    g.act = A8_ACT_MUL_GELU_GRAD; g.zin = e->z1; g.ldz = e->dff; g.colsum = e->db1;
    if (int rc = a8_gemm(&g, stream)) return rc;
    // a8_set_tuning key 23 = 1 (round-5 experiment, off): dx1 = dz1 W1 + dx1_res in the product's epilogue, so that the LayerNorm backward
    // below reads ONE gradient tensor instead of two -- one 9.6 MB stream less in a latency-bound kernel, and no gain in the step
    // (11.713 vs 11.677 ms, profiles/r05_step_ab.txt): the epilogue's second input costs the product what the LayerNorm saves
    g = nt_desc(e->dtype, e->dz1, e->w1_t, e->dx1_ffn, M, d, e->dff, nullptr);
    const bool fuse_res = g_a8_tuning[23] == 1;
    if (fuse_res) { g.act = A8_ACT_ADD; g.zin = e->dx1_res; g.ldz = d; }
    if (int rc = a8_gemm(&g, stream)) return rc;
    if (int rc = a8i_layernorm_bwd(e->dtype, fuse_res ? e->dx1_ffn : e->dx1_res, fuse_res ? nullptr : e->dx1_ffn, e->ao, e->x, e->g2, e->mean2, e->rstd2, e->dao, drop ? e->dx_res : nullptr,
                                   e->dg2, e->dbe2, e->dbo, M, d, e->p, e->seed1, 3, rng0, stream, e->ln2_partials))
        return rc;
    g = nt_desc(e->dtype, e->dao, e->wo_t, e->dout_o, M, d, d, nullptr);
    if (int rc = a8_gemm(&g, stream)) return rc;
    const char* q = (const char*)e->qkv;
    char* dq = (char*)e->dqkv;
    return a8i_attn_bwd(e->dtype, q, q + (size_t)d * esz, q + (size_t)2 * d * esz, e->o, e->dout_o, e->lse, e->keymask, dq, dq + (size_t)d * esz,
                        dq + (size_t)2 * d * esz, e->delta, nullptr, e->B, e->T, e->H, 3 * d, d, 3 * d, 1.0f / std::sqrt((float)(d / e->H)), e->p,
                        e->seed_attn, e->b0 * e->H, stream);
}

// The whole backward of one encoder layer behind ONE C call (include/a8hip.h): the host-side form of what audio8_amd/ops.py's
// EncoderLayerFn._backward issues as ~10 ctypes calls, two stream events and a dozen Python closures per layer.
namespace {
struct EventPair { hipEvent_t e[2]; bool ok = false; };
EventPair& layer_events() {
    static thread_local EventPair p;
    if (!p.ok) p.ok = hipEventCreateWithFlags(&p.e[0], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&p.e[1], hipEventDisableTiming) == hipSuccess;
    return p;
}
a8_gemm_desc tn_desc(int dtype, const void* dy, const void* x, float* out, int n_out, int n_in, int m, int accumulate) {
    a8_gemm_desc d;
    std::memset(&d, 0, sizeof(d));
    d.size = (int)sizeof(d);
    d.dtype = dtype; d.layout = A8_GEMM_TN;
    d.M = n_out; d.N = n_in; d.K = m;
    d.A = dy; d.lda = n_out;
    d.B = x; d.ldb = n_in;
    d.C = out; d.ldc = n_in;
    d.ldz = n_in;
    d.out_f32 = 1; d.accumulate = accumulate; d.splitk = 1; d.alpha = 1.0f;
    return d;
}
}  // namespace

extern "C" int a8_encoder_layer_bwd_full(const a8_encoder_layer_bwd_full_desc* f, void* main_stream, void* aux_stream, void* side_stream) {
    A8_REQUIRE(f != nullptr, "a8_encoder_layer_bwd_full: null descriptor");
    A8_REQUIRE(f->size == (int)sizeof(*f), "a8_encoder_layer_bwd_full: descriptor size %d, expected %d", f->size, (int)sizeof(*f));
    A8_REQUIRE(f->nhalves == 1 || f->nhalves == 2, "a8_encoder_layer_bwd_full: nhalves = %d", f->nhalves);
    A8_REQUIRE(f->nhalves == 1 || aux_stream != main_stream, "a8_encoder_layer_bwd_full: two ranges need two streams");
    A8_REQUIRE(side_stream != nullptr && side_stream != main_stream, "a8_encoder_layer_bwd_full: the weight gradients need their own stream");
    const a8_encoder_layer_bwd_desc& h0 = f->half[0];
    const int d = h0.d, dff = h0.dff, dtype = h0.dtype;
    int rows = 0;
    for (int i = 0; i < f->nhalves; ++i) {
        A8_REQUIRE(f->half[i].d == d && f->half[i].dff == dff && f->half[i].dtype == dtype && f->half[i].T == h0.T, "a8_encoder_layer_bwd_full: the ranges differ in shape");
        rows += f->half[i].B * f->half[i].T;
    }
    A8_REQUIRE(rows == f->M, "a8_encoder_layer_bwd_full: M = %d, the ranges hold %d rows", f->M, rows);
    A8_REQUIRE((f->ln1_ws == nullptr) == (f->ln2_ws == nullptr), "a8_encoder_layer_bwd_full: ln1_ws and ln2_ws go together");
    EventPair& ev = layer_events();
    if (!ev.ok) return a8_set_error(A8_ERR_LAUNCH, "a8_encoder_layer_bwd_full: could not create its events");
    hipStream_t sm = (hipStream_t)main_stream, sa = (hipStream_t)aux_stream, ss = (hipStream_t)side_stream;
    // 1. the data-gradient chains
    if (int rc = a8_encoder_layer_bwd(&f->half[0], main_stream)) return rc;
    if (f->nhalves == 2)
        if (int rc = a8_encoder_layer_bwd(&f->half[1], aux_stream)) return rc;
    // 2. the weight-gradient stream follows both
    bool okev = hipEventRecord(ev.e[0], sm) == hipSuccess && hipStreamWaitEvent(ss, ev.e[0], 0) == hipSuccess;
    if (f->nhalves == 2) okev = okev && hipEventRecord(ev.e[1], sa) == hipSuccess && hipStreamWaitEvent(ss, ev.e[1], 0) == hipSuccess;
    if (!okev) return a8_set_error(A8_ERR_LAUNCH, "a8_encoder_layer_bwd_full: stream ordering failed: %s", hipGetErrorString(hipGetLastError()));
    // 3. weight gradients (one grouped launch: df^T h, dz1^T x1, dao^T o, dqkv^T x), then the small column-sum work
    a8_gemm_desc g[4] = {tn_desc(dtype, h0.df, f->h, f->dw2, d, dff, f->M, f->acc_w2), tn_desc(dtype, h0.dz1, h0.x1, f->dw1, dff, d, f->M, f->acc_w1),
                         tn_desc(dtype, h0.dao, h0.o, f->dwo, d, d, f->M, f->acc_wo), tn_desc(dtype, h0.dqkv, h0.x, f->dwqkv, 3 * d, d, f->M, f->acc_wqkv)};
    if (int rc = a8_gemm_grouped(g, 4, side_stream)) return rc;
    if (f->ln1_ws) {
        if (int rc = a8_colsum_partials_reduce(f->ln1_ws, f->ln_rows, d, h0.dg1, h0.dbe1, h0.db2, side_stream)) return rc;
        if (int rc = a8_colsum_partials_reduce(f->ln2_ws, f->ln_rows, d, h0.dg2, h0.dbe2, h0.dbo, side_stream)) return rc;
    }
    if (f->dbqkv)
        if (int rc = a8_gelu_bwd_bias(dtype, h0.dqkv, nullptr, nullptr, f->dbqkv, f->M, 3 * d, side_stream)) return rc;
    // 4. the QKV data gradient (+ the residual branch's gradient), each range on its stream
    const int esz = dtype == A8_BF16 ? 2 : 4;
    int r0 = 0;
    for (int i = 0; i < f->nhalves; ++i) {
        const a8_encoder_layer_bwd_desc& hi = f->half[i];
        const int m = hi.B * hi.T;
        a8_gemm_desc q = nt_desc(dtype, hi.dqkv, f->wqkv_t, (char*)f->dx + (size_t)r0 * d * esz, m, d, 3 * d, nullptr);
        q.act = A8_ACT_ADD; q.zin = hi.dx_res; q.ldz = d;
        if (int rc = a8_gemm(&q, i == 0 ? main_stream : aux_stream)) return rc;
        r0 += m;
    }
    return A8_OK;
}
