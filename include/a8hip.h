/* a8hip -- C ABI of the MI355X-native wav2vec2 hot path (drop-in for mead-ml/audio8's
 * audio8/wav2vec2.py + audio8/ctc.py forward/backward).
 *
 * The reference has no FFI of its own (it is pure Python over ATen); this header is the
 * boundary a maintainer binds with ctypes.  Every entry point takes raw device
 * pointers, explicit sizes, a dtype enum and the HIP stream to launch on, and returns 0 on success
 * or a non-zero status whose text a8_last_error() gives.  No entry point allocates, synchronises or
 * aborts; a failed call leaves the stream usable (audio8/train.py:294,400 catches per-step errors
 * and continues, so must we).
 *
 * Activation layout is channels-last everywhere: [B, T, C] row-major ("rows" = B*T).  dtype is the
 * activation type (A8_BF16 or A8_F32); statistics, losses, weight gradients and optimizer state are
 * always f32.  Each declaration cites the reference code it replaces.
 */
#ifndef A8HIP_H
#define A8HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a descriptor struct's layout or an entry point's meaning changes.  A binding built against another value must refuse
 * to load (audio8_amd/_lib.py checks a8_version() == the value it was written for).  Since 120 every descriptor struct also starts with
 * `size` = sizeof(the struct as the CALLER compiled it): an entry point returns A8_ERR_ARG instead of reading a shorter or longer struct
 * than its own (120: size fields; a8_ctc_fwd_bwd's scratch grew to A8_CTC_SCRATCH_FLOATS; a8_get_tuning, a8_delay_us,
 * a8_encoder_layer_bwd_full). */
#define A8HIP_VERSION 120

enum { A8_DT_F32 = 0, A8_DT_BF16 = 1 };
enum { A8_GEMM_NT = 0, A8_GEMM_NN = 1, A8_GEMM_TN = 2 };
enum { A8_ACT_NONE = 0, A8_ACT_GELU = 1, A8_ACT_MUL_GELU_GRAD = 2, A8_ACT_ADD = 3 };

int a8_version(void);
const char* a8_last_error(void);      /* thread-local text of the last non-zero status */

/* ---- dense contraction: every Linear / Dense / strided Conv1d(512->512) of the path -------------
 * Replaces F.linear inside eight_mile Dense (audio8/wav2vec2.py:909,916-917,684,762), the q/k/v/o and
 * FFN projections of TransformerEncoderStack (:613-622) and, through the row remap, nn.Conv1d layers
 * 1..6 of ConvFeatureExtractionModel (:425-429).
 * C[M,N] = epi(alpha * A . B^T + bias);  layout NT: A[M][K],B[N][K]; NN: A[M][K],B[K][N]; TN: A[K][M],B[K][N].
 * Outer index o of an operand addresses (o / rpb) * bs + (o % rpb) * ld elements (rpb = 0: o * ld).
 * Aliasing rule: C (and zout) may alias zin element for element (in-place "+ zin" / "x gelu'(zin)": every element is then read and
 * written once, by the same lane); they must not otherwise overlap A, B or zin.  The library sends an aliased problem to the
 * 128 x 128 kernels (the stream kernel's shifted last tile stores some rows twice). */
typedef struct a8_gemm_desc {
    int size;                             /* sizeof(a8_gemm_desc) */
    int dtype, layout;
    int M, N, K;
    const void* A; int64_t lda, a_rpb, a_bs;
    const void* B; int64_t ldb, b_rpb, b_bs;
    void* C;       int64_t ldc, c_rpb, c_bs;
    const float* bias;  int bias_rows;   /* bias[N] (or bias[M] when bias_rows) */
    int act;                              /* A8_ACT_* ; MUL_GELU_GRAD multiplies by gelu'(zin); ADD adds zin (residual gradient) */
    const void* zin; void* zout; int64_t ldz, z_rpb, z_bs;   /* zout: pre-activation copy (dtype); z rows use their own remap */
    int out_f32;                          /* C is f32 regardless of dtype */
    int accumulate;                       /* C += (f32 atomics) */
    int splitk;                           /* >1: split the K range over workgroups (f32 atomics) */
    float alpha;
    float* colsum;                        /* optional f32 [N] += column sums of the final C values (a bias gradient riding on
                                             the data-gradient GEMM that produces dz); not with split-K */
} a8_gemm_desc;
int a8_gemm(const a8_gemm_desc* d, void* stream);
/* n (<= 8) independent problems of one dtype and layout in ONE launch (block ranges partition the grid): the four
 * weight-gradient products of an encoder layer fill the chip together without split-K.  Replaces the same
 * per-parameter autograd GEMMs as a8_gemm. */
int a8_gemm_grouped(const a8_gemm_desc* descs, int n, void* stream);
/* developer tuning knobs, never needed for correctness (defaults are the measured best): key 1 LayerNorm-backward geometry
 * (waves*100 + rows per wave); 2..4 attention fwd / dK,dV / dQ launch geometry (2: < 10 = the 32 x 32-MFMA forward kernel, default;
 * 3 / 4: 41 = the 16 x 16 backward kernels, default; 32 = the round-6 32 x 32-MFMA backward kernels, tested and slower); 5 minimum M for the
 * stream bf16 NT GEMM (N % 256 == 0, K % 128 == 0: BM x 256 tiles on persistent blocks, the conv layers' forward; default 20000, 0 = off),
 * 7 its forced tile height (160 / 192 / 224 / 256; 0 = chosen per launch), 8 its grid in CUs' worths of persistent blocks (default 1: all
 * resident from the start; 2..4: shorter static shares, surplus blocks dispatched as CUs free up -- +0.09 ms alone on the GPU, for runs where
 * another kernel holds CUs at launch); 6 positional-conv weight-gradient loop splits; 17 LayerNorm-backward geometry of the partial-row form (default 801);
 * 9 = 1: conv layer 0 on the VALU in bf16 mode too; 14 asymmetric LDS ring for bf16 NT GEMMs (A requested two k-tiles ahead, B one):
 * 0 = off, otherwise the minimum M (default 1000: the conv layers and the encoder's products -- issued as half-batches on two streams,
 * most of their blocks sit alone on a CU, where the deeper A ring hides what a co-resident block otherwise would); 15 the same ring for
 * bf16 TN products (the weight gradients; default 1 = on); 19 = 1: the stream kernel also takes plain (not row-remapped) problems
 * (default 0: only the conv layers' remapped products -- a persistent block owns its CU, which costs the encoder's projections their
 * co-running streams); 20 / 21 AdamW grid / cache policy; 22 = 1: LDS-staged epilogue in the 128 x 128 NT kernels; 23 = 1: residual add in the
 * FFN data gradient's epilogue; 25 = 1: predicate-form operand addressing in the 128 x 128 NT kernels. */
int a8_set_tuning(int key, int value);
/* The knobs are PROCESS-WIDE (one table per loaded library): they choose among kernel variants with bit-identical results, so models or
 * threads sharing a process can only affect each other's speed.  Reads and writes are atomic; a change applies to launches issued after
 * a8_set_tuning returns.  a8_get_tuning reads a key back (so that a caller can restore what it found). */
int a8_get_tuning(int key, int* value);
/* Measurement helper: one wave that holds `stream` for `us` microseconds of wall clock -- the stand-in for a collective's transport
 * time when the exposed (un-overlapped) part of the gradient exchange is measured on one GPU (audio8/pretrain.py:152-159,179: DDP's
 * bucketed all-reduce under backward).  No memory traffic, one wave slot. */
int a8_delay_us(float us, void* stream);

/* ---- per-step weight preparation (the f32 -> compute-dtype casts the reference's autocast-free f32 path never needs;
 * conv kernels [rows][cols/kw][kw] are re-laid-out to [rows][kk*(cols/kw) + ci]).  dst [rows][cols] and/or dstT [cols][rows]. */
typedef struct a8_wprep_desc {
    const float* src; void* dst; void* dstT;
    int rows, cols, kw;      /* logical [rows][cols]; kw > 1: src is a conv kernel [rows][cols/kw][kw] */
    int tap;                 /* -1: all taps, column c = kk*(cols/kw)+ci; >= 0: single tap, src[(r*cols + c)*kw + tap] */
    int ld_dst, ld_dstT;     /* leading dimensions of dst / dstT (0: dense) */
} a8_wprep_desc;
int a8_prepare_weights(int dtype, const a8_wprep_desc* descs /* host array */, int n, void* stream);
/* one-launch form: descs and tile_start (n+1 ints, prefix sums of ceil(rows/64)*ceil(cols/64)) in device memory */
int a8_prepare_weights_device(int dtype, const a8_wprep_desc* descs_dev, const int* tile_start_dev, int n, int total_tiles, void* stream);

/* ---- conv layer 0 + GroupNorm + GELU (audio8/wav2vec2.py:417-423, Cin=1, k=10, s=5) --------------
 * x f32 [B,L]; w f32 [C,10]; out dtype [B,L0,C] channels-last; stats f32 [B,C,2] = (mean, rstd).
 * part is scratch, 8-byte aligned, A8_CONV0_FWD_SCRATCH_FLOATS(B, L0) floats (f64 waveform moments per 512-frame chunk: the
 * GroupNorm statistics come from 65 channel-free sums per utterance, not from a pass over the conv outputs). */
#define A8_CONV0_FWD_SCRATCH_FLOATS(B, L0) ((size_t)(B) * (((size_t)(L0) + 511) / 512) * 65 * 2)
int a8_conv0_gn_gelu_fwd(int dtype, const float* x, const float* w, const float* gn_w, const float* gn_b,
                         void* out, float* stats, float* part, int B, int L, int C, float eps, void* stream);
/* backward: dout dtype [B,L0,C] -> dw f32 [C,10], dgn_w, dgn_b f32 [C] (all accumulated, +=).
 * red is scratch of A8_CONV0_BWD_SCRATCH_FLOATS(B) floats, 8-byte aligned (zeroed by the call). */
#define A8_CONV0_BWD_SCRATCH_FLOATS(B) ((size_t)(B) * (2 * 65 + 512 * 12))
int a8_conv0_gn_gelu_bwd(int dtype, const float* x, const float* w, const float* gn_w, const float* gn_b,
                         const float* stats, const void* dout, float* dw, float* dgn_w, float* dgn_b,
                         float* red, int B, int L, int C, void* stream);

/* ---- strided conv dgrad scatter (col2im) fused with GELU' (layers 1..6 backward) ----------------
 * dcol dtype [B*Lout, k*C] (one GEMM row per output frame) -> dx dtype [B,Lin,C]:
 * dx[b,p,c] = gelu'(z[b,p,c]) * sum_{t*stride+kk = p} dcol[b,t,kk,c]   (z = pre-activation of the layer below;
 * z == NULL: no GELU' factor) */
int a8_col2im_gelu_grad(int dtype, const void* dcol, const void* z, void* dx, int B, int Lin, int Lout, int C,
                        int k, int stride, void* stream);

/* ---- LayerNorm (nn.LayerNorm audio8/wav2vec2.py:904,623 and eight_mile ln1/ln2) ------------------
 * y = LN(res + dropout(x)) * gamma + beta over the last dim; res may be NULL; p = 0 disables dropout.
 * mean/rstd f32 [rows] are saved for backward. */
int a8_layernorm_fwd(int dtype, const void* x, const void* res, const float* gamma, const float* beta, void* y,
                     float* mean, float* rstd, int rows, int cols, float eps, float p, uint32_t seed, uint32_t rng_stream,
                     void* stream);
/* dy (+ dy2 if non-NULL) -> ds = d(res + dropout(x)); writes dres = ds (if dres) and dx = ds * keep/(1-p) (if dx);
 * dgamma/dbeta f32 [cols] accumulated (+=; both NULL to skip); dbias (optional) f32 [cols] += column sums of dx, the
 * bias gradient of the Linear that produced x (eight_mile Dense bias, audio8/wav2vec2.py:613-622 block). */
int a8_layernorm_bwd(int dtype, const void* dy, const void* dy2, const void* x, const void* res, const float* gamma,
                     const float* mean, const float* rstd, void* dx, void* dres, float* dgamma, float* dbeta, float* dbias,
                     int rows, int cols, float p, uint32_t seed, uint32_t rng_stream, void* stream);

/* ---- element-wise helpers ----------------------------------------------------------------------- */
int a8_cast(int src_dtype, int dst_dtype, const void* src, void* dst, int64_t n, void* stream);
/* y = x * keep / (1-p); the same call with the same (seed, rng_stream) is its own backward. (nn.Dropout) */
int a8_dropout(int dtype, const void* x, void* y, int64_t n, float p, uint32_t seed, uint32_t rng_stream, void* stream);
/* features[time_mask] = mask_emb (audio8/wav2vec2.py:939,717): x dtype [rows,cols] in place, mask u8[rows], emb f32[cols] */
int a8_mask_fill_rows(int dtype, void* x, const uint8_t* mask, const float* emb, int rows, int cols, void* stream);
/* backward: demb[cols] += sum over masked rows of dy; dy[masked rows] = 0 (in place) */
int a8_mask_fill_rows_bwd(int dtype, void* dy, const uint8_t* mask, float* demb, int rows, int cols, void* stream);
/* x[:, c] = 0 where cmask[b, c]  (channel mask, audio8/wav2vec2.py:718-721); cmask u8 [B, cols]; same call is its backward */
int a8_mask_zero_cols(int dtype, void* x, const uint8_t* cmask, int B, int T, int cols, void* stream);
/* x[row] = 0 where !keep[row]  (x[~pad_mask] = 0, audio8/wav2vec2.py:631-632) */
int a8_zero_rows(int dtype, void* x, const uint8_t* keep, int rows, int cols, void* stream);
/* dst[i] = src[idx[i]] (masked gather :946) / dst[idx[i]] += src[i] (its backward; dst dtype, unique idx) */
int a8_gather_rows(int dtype, const void* src, const int32_t* idx, void* dst, int n, int cols, void* stream);
int a8_scatter_add_rows(int dtype, const void* src, const int32_t* idx, void* dst, int n, int cols, void* stream);
/* y = a + b */
int a8_add(int dtype, const void* a, const void* b, void* y, int64_t n, void* stream);

/* ---- positional conv (eight_mile Conv1DSame(d,d,k=128,groups=16)+GELU, audio8/wav2vec2.py:600-609,634-638)
 * x dtype [B,T,d]; w dtype [G][d/G out][k][d/G in] (weight-normed, re-laid-out by the host); bias f32[d].
 * add_residual = 1: y = x + gelu(conv(x) + bias); 0: y = gelu(conv(x) + bias); -1: y = conv(x) (data-gradient mode).
 * z (pre-GELU, dtype, may be NULL) is saved for backward.
 * The same kernel computes the data gradient when given the flipped/transposed weights and pads swapped. */
int a8_posconv_fwd(int dtype, const void* x, const void* w, const float* bias, void* y, void* z,
                   int B, int T, int d, int groups, int k, int pad_left, int add_residual, void* stream);
/* dw f32 [G][dg_out][k][dg_in] += sum_{b,t} dz[b,t,co] x[b,t+kk-pad_left,ci] */
int a8_posconv_wgrad(int dtype, const void* x, const void* dz, float* dw, int B, int T, int d, int groups, int k,
                     int pad_left, void* stream);
/* Weight norm of the positional conv kernel (nn.utils.weight_norm(name="weight", dim=2), audio8/wav2vec2.py:609):
 * v f32 [d][d/groups][k], g f32 [k]:  w[o][ci][kk] = g[kk] * v[o][ci][kk] / ||v[:, :, kk]||.
 * a8_weightnorm_pack writes the two operand images a8_posconv_fwd takes, in `dtype`:
 *   wf [G][co][kk][ci] = w[g*dg+co][ci][kk] (forward),  wt [G][ci][kk'][co] = w[g*dg+co][ci][k-1-kk'] (data gradient),
 * and inv_norm f32 [k] = 1 / ||v[:, :, kk]|| for the backward pass.  scratch: f32 [d*k].  k: a power of two in 16..128.
 * a8_weightnorm_bwd takes dwg f32 [G][co][kk][ci] (what a8_posconv_wgrad produces) through the weight norm:
 *   dgain[kk] += <dw, v>_kk * inv_norm[kk];   dv += g * inv_norm * (dw - v * <dw, v>_kk * inv_norm^2).   scratch: f32 [d*k + k]. */
int a8_weightnorm_pack(int dtype, const float* v, const float* g, void* wf, void* wt, float* inv_norm, float* scratch,
                       int d, int groups, int k, void* stream);
int a8_weightnorm_bwd(const float* dwg, const float* v, const float* g, const float* inv_norm, float* dv, float* dgain,
                      float* scratch, int d, int groups, int k, void* stream);
/* dz = dy * gelu'(z) (z NULL: dz = dy); dbias[cols] += column sums of dz.  dz and dbias may each be NULL. */
int a8_gelu_bwd_bias(int dtype, const void* dy, const void* z, void* dz, float* dbias, int rows, int cols, void* stream);

/* ---- multi-head self-attention core (eight_mile SeqScaledDotProductAttention) ----
 * q,k,v,o dtype, element (b,t,h,c) at ((b*T + t) * ld + h*64 + c); head dim is 64.
 * keymask u8 [B,T] (1 = attend) or NULL; masked scores are filled with -1e9 before softmax as the reference does.
 * dropout on the attention probabilities with keep-prob 1-p (p = 0: off).  lse f32 [B,H,T] saved for backward. */
int a8_attn_fwd(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* keymask,
                int B, int T, int H, int64_t ld_qkv, int64_t ld_o, float scale, float p, uint32_t seed, void* stream);
/* delta is scratch f32 [B,H,T] (rowsum(dO * O): written by the dQ kernel for its own rows, read by the dK/dV kernel).
 * dbias (optional, f32 [3][H*64]) += column sums of dq | dk | dv: the gradients of the q / k / v projection biases
 * (eight_mile MultiHeadedAttention w_Q/w_K/w_V), one atomic per column per block -- no separate pass over dqkv. */
int a8_attn_bwd(int dtype, const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                const uint8_t* keymask, void* dq, void* dk, void* dv, float* delta, float* dbias, int B, int T, int H, int64_t ld_qkv,
                int64_t ld_o, int64_t ld_dqkv, float scale, float p, uint32_t seed, void* stream);

/* ---- one encoder layer forward as ONE call (the launch sequence of eight_mile's post-LN TransformerEncoder, SURVEY 8a row a5:
 * QKV GEMM, attention, output projection, LN2(x + drop), FFN1 + GELU, FFN2, LN1(x1 + drop)); replaces nothing new -- it issues the
 * entry points above in order -- but costs the host one C transition instead of seven.  All buffers are the caller's:
 * x, qkv [M,3d], o, ao, x1, f, x2 [M,d], z1, h [M,dff] in dtype; lse f32 [B,H,T]; mean / rstd f32 [M]; weights in dtype, row-major
 * [out,in] (wqkv = [3d,d]); biases and LayerNorm parameters f32.  Seeds as a8_attn_fwd / a8_layernorm_fwd take them (p = 0: unused). */
typedef struct a8_encoder_layer_desc {
    int size;                /* sizeof(a8_encoder_layer_desc) */
    int dtype, B, T, H, d, dff;
    float p, eps;
    uint32_t seed_attn, seed1, seed2;
    int b0;                  /* 0, or the index of this call's first utterance in the logical batch: a call on utterances [b0, b0 + B) (all pointers
                                already advanced to them) draws the dropout masks the whole-batch call would draw there -- two such calls on two
                                streams overlap each other's under-filled launches */
    const void* x; const uint8_t* keymask;
    const void *wqkv, *wo, *w1, *w2;
    const float *bqkv, *bo, *b1, *b2, *g1, *be1, *g2, *be2;
    void *qkv, *o, *ao, *x1, *z1, *h, *f, *x2;
    float *lse, *mean2, *rstd2, *mean1, *rstd1;
} a8_encoder_layer_desc;
int a8_encoder_layer_fwd(const a8_encoder_layer_desc* e, void* stream);

/* The data-gradient chain of that layer's backward up to and including the attention core, as one call (LN1 backward carrying the b2
 * gradient, the FFN2 data gradient x gelu'(z1) carrying the b1 gradient, the FFN1 data gradient, LN2 backward carrying the bo gradient,
 * the output projection's data gradient, attention backward).  dx2 = gradient of the layer output; x .. rstd1 as saved by the forward
 * call; w2_t [dff,d], w1_t [d,dff], wo_t [d,d] = the TRANSPOSED working copies (a8_prepare_weights); dg1 .. dbo f32, accumulated (+=).
 * Written: df, dz1, dx1_ffn, dao, dout_o, dqkv [M,3d], delta f32 [B,H,T], and with p > 0 dx1_res, dx_res (p = 0: pass df / dao again).
 * The weight gradients and the QKV data gradient (dx = dqkv Wqkv + dx_res) are the caller's. */
typedef struct a8_encoder_layer_bwd_desc {
    int size;                /* sizeof(a8_encoder_layer_bwd_desc) */
    int dtype, B, T, H, d, dff;
    float p;
    uint32_t seed_attn, seed1, seed2;
    int b0;                  /* as in a8_encoder_layer_desc: first utterance of this call in the logical batch (dropout counters) */
    const void *dx2, *x, *qkv, *o, *ao, *x1, *z1, *f; const uint8_t* keymask;
    const float *lse, *mean2, *rstd2, *mean1, *rstd1, *g1, *g2;
    const void *w2_t, *w1_t, *wo_t;
    float *dg1, *dbe1, *dg2, *dbe2, *db2, *db1, *dbo;
    void *df, *dx1_res, *dz1, *dx1_ffn, *dao, *dx_res, *dout_o, *dqkv;
    float* delta;
    float *ln1_partials, *ln2_partials;   /* NULL: dg / dbe / db2 / dbo are accumulated by the kernels (f32 atomics).  Otherwise f32
                                             [a8_layernorm_bwd_partial_rows(B*T)][3][d] each: per-block partial column sums (dgamma | dbeta |
                                             bias gradient), to be added into dg1, dbe1, db2 (ln1) / dg2, dbe2, dbo (ln2) by
                                             a8_colsum_partials_reduce on any stream ordered behind this call */
} a8_encoder_layer_bwd_desc;
int a8_encoder_layer_bwd(const a8_encoder_layer_bwd_desc* e, void* stream);
/* The WHOLE backward of that layer as one call (round 6: the host's share -- one Python thread per GPU issues the step, audio8/pretrain.py:171-186):
 * the chains of one or two utterance ranges (half[0] on `main`, half[1] on `aux`), then -- on `side`, behind both -- the four weight
 * gradients as one grouped launch, the LayerNorm / bias partial-row reductions and the q / k / v bias column sums, and the QKV data
 * gradient dx = dqkv Wqkv + dx_res per range on its own stream.  Issues exactly the launches of a8_encoder_layer_bwd + a8_gemm_grouped +
 * a8_colsum_partials_reduce + a8_gelu_bwd_bias + a8_gemm in the order audio8_amd/ops.py issues them; stream ordering by two events of its
 * own per call.  half[i] as for a8_encoder_layer_bwd (rows of range i); half[0] starts at the batch's first row, the ranges are adjacent. */
typedef struct a8_encoder_layer_bwd_full_desc {
    int size;                                   /* sizeof(a8_encoder_layer_bwd_full_desc) */
    int nhalves;                                /* 1: the whole batch on `main`; 2: half[1] on `aux` */
    a8_encoder_layer_bwd_desc half[2];
    int M;                                      /* rows of the whole batch */
    const void* h;                              /* [M, dff] FFN activations (operand of the w_2 weight gradient) */
    const void* wqkv_t;                         /* [d, 3d] transposed working copy of the fused QKV weight */
    void* dx;                                   /* out [M, d] */
    float *dw2, *dw1, *dwo, *dwqkv;             /* f32 [d,dff], [dff,d], [d,d], [3d,d] */
    int acc_w2, acc_w1, acc_wo, acc_wqkv;       /* 1: accumulate (f32 atomics); 0: first gradient into a clean slot, stored */
    float* dbqkv;                               /* f32 [3d] += column sums of dqkv */
    const float *ln1_ws, *ln2_ws; int ln_rows;  /* the partial rows of both ranges (what half[i].ln*_partials point into), contiguous; or NULL */
} a8_encoder_layer_bwd_full_desc;
int a8_encoder_layer_bwd_full(const a8_encoder_layer_bwd_full_desc* f, void* main_stream, void* aux_stream, void* side_stream);
/* LayerNorm backward without same-address atomics (audio8/wav2vec2.py:613-623 LayerNorms' weight / bias gradients): the number of
 * partial rows one backward launch over `rows` rows writes, and the reduction out_k[c] += sum_b ws[b][k][c] (ws f32 [nrows][3][cols]; a NULL
 * out_k is skipped).  391 x 2304 same-address atomics per launch forced few, fat, latency-exposed blocks: -0.06 ms per step, and the sums are
 * deterministic up to the 16 row stripes of the reduction. */
int a8_layernorm_bwd_partial_rows(int rows);
int a8_colsum_partials_reduce(const float* ws, int nrows, int cols, float* out0, float* out1, float* out2, void* stream);

/* ---- small runtime helpers of the step (no ATen kernels between the first conv launch and the optimizer) ------------------
 * a8_zero: bytes of device memory := 0 on the stream (hipMemsetAsync).
 * a8_conv_wgrad_fold: the strided-conv weight gradient leaves its split-K GEMM as dw2 f32 [co][k][cin] (reduction index kk*cin + ci);
 * the parameter (nn.Conv1d weight, audio8/wav2vec2.py:425-429) is [co][cin][k]:  grad[o][c][kk] += dw2[o][kk][c], and dw2 := 0 behind the
 * read (the caller keeps dw2 across steps: no fill launch). */
int a8_zero(void* p, int64_t bytes, void* stream);
int a8_conv_wgrad_fold(float* dw2, float* grad, int co, int cin, int k, void* stream);

/* ---- Gumbel-softmax product quantiser (GumbelVectorQuantizer.forward audio8/wav2vec2.py:547-576) ----
 * logits f32 [R, V] (R = B*Tm*G rows), gumbel f32 [R,V] or NULL; gumbel_gen f32 [R,V] or NULL: when gumbel is NULL and gumbel_gen
 * is not, the Gumbel(0,1) noise of F.gumbel_softmax is drawn inside the kernel from a counter hash of (seed, element) and written to
 * gumbel_gen (backward reads it); both NULL: eval, plain argmax;
 * codes i32 [R]; avg_probs f32 [V] = mean over rows of softmax(logits) (written, not accumulated);
 * scratch: f32 [A8_VQ_SELECT_SCRATCH_FLOATS(R, V)] (per-block partial rows: no atomics anywhere) */
#define A8_VQ_SELECT_SCRATCH_FLOATS(R, V) ((((size_t)(R) + 15) / 16) * (size_t)(V))
int a8_vq_select(const float* logits, const float* gumbel, float* gumbel_gen, uint32_t seed, float tau, int32_t* codes,
                 float* avg_probs, float* scratch, int R, int V, void* stream);
/* q[n, g*D + :] = vars[g*V + codes[n*G+g], :]  (codebook gather; vars f32 [G*V, D]); q dtype [N, G*D] */
int a8_vq_gather(int dtype, const float* vars, const int32_t* codes, void* q, int N, int G, int V, int D, void* stream);
/* prob_ppl = sum exp(-sum p log(p + 1e-7)) over avg_probs[V] -> ppl[0]; dppl_davg[V] */
int a8_vq_ppl(const float* avg_probs, float* ppl, float* dppl_davg, int V, void* stream);
/* Scalar composition of the pretraining loss (Wav2Vec2Loss.__call__, audio8/wav2vec2.py:390-392), on device so that the
 * step has no host read-back and no chain of one-element kernels:
 *   loss[0] = w_xe * (xe_sum[0] / rows) + w_div * (n_vars - ppl[0]) / n_vars
 * a8_loss_combine_bwd: dxe_sum[0] = dloss[0] * w_xe / rows;  dppl[0] = -dloss[0] * w_div / n_vars. */
int a8_loss_combine(const float* xe_sum, const float* ppl, int rows, float n_vars, float w_xe, float w_div, float* loss, void* stream);
int a8_loss_combine_bwd(const float* dloss, int rows, float n_vars, float w_xe, float w_div, float* dxe_sum, float* dppl, void* stream);
/* backward: dq dtype [N, G*D]; dvars f32 [G*V, D] += ; dlogits f32 [R,V] =
 *   straight-through soft path: d/dlogits of sum_v ysoft_v * <dq_g, vars_v>  (ysoft = softmax((logits+gumbel)/tau))
 * + diversity path: gppl * dppl/dlogits through avg_probs = mean softmax(logits). */
int a8_vq_bwd(int dtype, const float* logits, const float* gumbel, float tau, const int32_t* codes, const float* vars,
              const void* dq, const float* eproj /* f32 [R,V]: <dq_g, vars[g,v,:]>, from a8_gemm */, const float* dppl_davg, const float* gppl, float* dlogits, float* dvars,
              int N, int G, int V, int D, int training, void* stream);

/* ---- negatives gather + cosine + cross-entropy (Sampler.negatives + Wav2Vec2Loss, audio8/wav2vec2.py:371-392,955-976)
 * ctx dtype [B*Tm, C] (context at masked frames), lat dtype [B*Tm, C] (quantised latents), neg_idx i32 [B*Tm, N]
 * (row indices into lat).  logits f32 [B*Tm, N+1] (column 0 = positive), loss_sum f32[1] += sum_rows CE(row, 0). */
int a8_contrastive_fwd(int dtype, const void* ctx, const void* lat, const int32_t* neg_idx, float* logits,
                       float* loss_sum, float* inv_norms /* out f32 [2][rows]: 1/|ctx row|, 1/|lat row| */, int rows, int N, int C, void* stream);
/* dctx, dlat f32 [rows, C] (dlat accumulated with atomics: zero it first); gscale multiplies d(loss_sum).
 * rows_per_group > 0 declares that a row's negatives come from its own group of that many consecutive rows (one utterance,
 * as Sampler.negatives draws them): the latent gradient is then a small dense product per group instead of
 * global atomics per pair (any pair that does not obey it is still handled); 0 = no assumption.
 * inv_norms: what a8_contrastive_fwd wrote for the same rows, or NULL (recomputed).
 * scratch: A8_CONTRASTIVE_BWD_SCRATCH_FLOATS(rows, N, rows_per_group) floats, 16-byte aligned. */
#define A8_CONTRASTIVE_BWD_SCRATCH_FLOATS(rows, N, rows_per_group) \
    ((size_t)(rows) * (2 + 2 * ((N) + 1) + 1 + ((size_t)(rows_per_group) + 7) + 256 + 2 * 256) + 32)
int a8_contrastive_bwd(int dtype, const void* ctx, const void* lat, const int32_t* neg_idx, const float* logits,
                       const float* gscale, const float* inv_norms, float* dctx, float* dlat, float* scratch, int rows, int N, int C,
                       int rows_per_group, void* stream);

/* ---- CTC (F.ctc_loss native path, audio8/ctc.py:186-206) + log_softmax head (audio8/wav2vec2.py:770) ----
 * lp f32 [T,B,V] log-probs; targets i32 flat (PAD/EOS already stripped), tgt_off i32 [B+1]; nll f32 [B];
 * grad f32 [T,B,V] = d nll_b / d lp (zero_infinity applied); scratch: A8_CTC_SCRATCH_FLOATS(T, B, Smax) floats (the per-frame
 * normalised log2 alpha and beta of every utterance + a feasibility flag each).  Two launches: alpha and beta recurse concurrently
 * (one block each per utterance -- one wave up to 128 extended labels 2 Smax + 1, four waves meeting at one barrier per frame beyond --, the
 * log-probability rows staged through LDS ahead of the recurrence), then one wave per frame forms the posteriors and the gradient. */
#define A8_CTC_SCRATCH_FLOATS(T, B, Smax) (2 * (size_t)(B) * (size_t)(T) * (64 * ((2 * (size_t)(Smax) + 1 + 63) / 64)) + (size_t)(B))
/* targets i64 [B,S] padded with PAD / EOS (audio8/ctc.py:193-194 `targets.masked_select(pad_mask)`) -> flat i32 [<= B*S] and
 * tgt_off i32 [B+1] = exclusive prefix sum of target_lengths i32 [B]; everything on the device (the reference's masked_select
 * is a host sync).  Row b's labels land at tgt_off[b], cut at target_lengths[b]; status (optional, i32 [2]) = {rows whose count of
 * non-PAD/EOS entries differs from target_lengths[b], max(target_lengths)} for a deferred host check. */
int a8_ctc_pack_targets(const int64_t* targets, const int32_t* target_lengths, int32_t* flat, int32_t* tgt_off, int32_t* status,
                        int B, int S, int pad, int eos, void* stream);
int a8_ctc_fwd_bwd(const float* lp, const int32_t* targets, const int32_t* tgt_off, const int32_t* in_len,
                   float* nll, float* grad, float* scratch, int T, int B, int V, int Smax, int blank, int zero_infinity,
                   void* stream);
/* greedy decode for the metrics (audio8/ctc.py:156-159): per utterance argmax over labels of frames [0, min(in_len, T)),
 * collapse repeats, drop blanks.  lp element (b,t,v) at b*stride_b + t*stride_t + v.  tokens i32 [B,T] (first out_len[b]
 * entries of row b valid), out_len i32 [B]. */
int a8_ctc_greedy_decode(int dtype, const void* lp, const int32_t* in_len, int32_t* tokens, int32_t* out_len,
                         int B, int T, int V, int64_t stride_b, int64_t stride_t, int blank, void* stream);
/* log_softmax over the label axis (audio8/wav2vec2.py:770): x dtype [rows, ldx] (first `cols` columns of each row are the
 * logits; ldx > cols when the head's GEMM padded the label count), y f32 [rows, cols].  Backward writes dx dtype
 * [rows, ldx] with zeros in the padding columns.  Any number of labels (letters or BPE vocabularies, audio8/train.py:194-198). */
int a8_log_softmax_fwd(const void* x, int dtype, float* y, int rows, int cols, int ldx, void* stream);
int a8_log_softmax_bwd(const float* y, const float* dy, void* dx, int dtype, int rows, int cols, int ldx, void* stream);

/* ---- optimizer step either side of the path (clip_grad_norm_ + AdamW, audio8/pretrain.py:182-184) ----
 * a8_sumsq: out[0] += sum x^2 ;  a8_adamw: p,m,v updated in place with g * min(1, max_norm / (sqrt(gnorm_sq[0]) + 1e-6));
 * zero_grad != 0 also clears g behind the read (optimizer.zero_grad() of the next step, audio8/pretrain.py:177, for free).
 * grad_denom (device int64[1] or NULL): gradients are further divided by grad_denom[0] -- the all-reduced batch size of
 * audio8/train.py:320-323 (`scale_grads(num_gpus / batch_size)` after DDP's mean), read on the device: no host sync. */
int a8_sumsq(const float* x, int64_t n, float* out, void* stream);
int a8_adamw(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
             float weight_decay, int step, const float* gnorm_sq, float max_norm, float grad_scale, const int64_t* grad_denom,
             int zero_grad, void* stream);

/* ---- host side of the boundary (no GPU work) ----
 * Negative indices of the contrastive loss exactly as the reference draws them: np.random.randint(0, T - 1, (B, n_neg * T)), the
 * ">= own frame -> +1" shift and the per-utterance row offset (audio8/wav2vec2.py:955-976), on a copy of numpy's global MT19937 state
 * (np.random.get_state(): key[624], pos) that is advanced exactly as numpy would advance it.  out: int32 [B, n_neg * T] flat row
 * indices into [B * T, C].  T is the number of masked frames per utterance. */
int a8_host_negative_indices(uint32_t* key, int* pos, int B, int T, int n_neg, int32_t* out);

#ifdef __cplusplus
}
#endif
#endif
