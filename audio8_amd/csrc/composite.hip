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
    A8_REQUIRE(e->B > 0 && e->T > 0 && e->H > 0 && e->d == e->H * 64 && e->dff > 0, "a8_encoder_layer_fwd: bad sizes B=%d T=%d H=%d d=%d dff=%d",
               e->B, e->T, e->H, e->d, e->dff);
    const int M = e->B * e->T, d = e->d, esz = e->dtype == A8_BF16 ? 2 : 4;
    a8_gemm_desc g = nt_desc(e->dtype, e->x, e->wqkv, e->qkv, M, 3 * d, d, e->bqkv);
    if (int rc = a8_gemm(&g, stream)) return rc;
    const char* q = (const char*)e->qkv;
    if (int rc = a8_attn_fwd(e->dtype, q, q + (size_t)d * esz, q + (size_t)2 * d * esz, e->o, e->lse, e->keymask, e->B, e->T, e->H, 3 * d, d,
                             1.0f / std::sqrt((float)(d / e->H)), e->p, e->seed_attn, stream))
        return rc;
    g = nt_desc(e->dtype, e->o, e->wo, e->ao, M, d, d, e->bo);
    if (int rc = a8_gemm(&g, stream)) return rc;
    if (int rc = a8_layernorm_fwd(e->dtype, e->ao, e->x, e->g2, e->be2, e->x1, e->mean2, e->rstd2, M, d, e->eps, e->p, e->seed1, 3, stream)) return rc;
    g = nt_desc(e->dtype, e->x1, e->w1, e->h, M, e->dff, d, e->b1);
    g.act = A8_ACT_GELU; g.zout = e->z1;
    if (int rc = a8_gemm(&g, stream)) return rc;
    g = nt_desc(e->dtype, e->h, e->w2, e->f, M, d, e->dff, e->b2);
    if (int rc = a8_gemm(&g, stream)) return rc;
    return a8_layernorm_fwd(e->dtype, e->f, e->x1, e->g1, e->be1, e->x2, e->mean1, e->rstd1, M, d, e->eps, e->p, e->seed2, 4, stream);
}
