"""Thin, autograd-free Python wrappers over the C ABI (one function per entry point).

Tensors are torch CUDA tensors used only as device-memory handles; every wrapper launches on
torch's current HIP stream.  All arithmetic happens inside liba8hip.so.
"""
import ctypes as C

import torch

from . import _lib as L


# When set to a list, every a8_gemm launch is bracketed by HIP events on the launch stream and
# (start, end, algorithmic_flops, layout) is appended -- bench.py's live roofline measurement.
GEMM_PROFILE = None


def _c(t):
    if t is not None and not t.is_contiguous():
        raise RuntimeError("a8hip: expected a contiguous tensor")
    return t


def _gemm_desc(layout, A, B, M, N, K, *, out=None, out_dtype=None, lda=None, ldb=None, ldc=None, bias=None, bias_rows=False,
               act=L.ACT_NONE, zin=None, zout=None, ldz=None, accumulate=False, splitk=1, alpha=1.0,
               a_rpb=0, a_bs=0, b_rpb=0, b_bs=0, c_rpb=0, c_bs=0, z_rpb=0, z_bs=0, a_off=0, c_off=0, z_off=0, colsum=None):
    L.require_cuda(A, B, out, bias, zin, zout, colsum)
    dtype = L.dt(A)
    assert B.dtype == A.dtype
    if layout == L.NT:
        lda = lda if lda is not None else K
        ldb = ldb if ldb is not None else K
    elif layout == L.NN:
        lda = lda if lda is not None else K
        ldb = ldb if ldb is not None else N
    else:
        lda = lda if lda is not None else M
        ldb = ldb if ldb is not None else N
    ldc = ldc if ldc is not None else N
    if out is None:
        out = torch.empty((M, N), device=A.device, dtype=out_dtype or A.dtype)
    esz = A.element_size()
    # one positional construction (field order of include/a8hip.h): a third of the host time of thirty attribute stores
    d = L.GemmDesc(L.GEMM_DESC_SIZE, dtype, layout, M, N, K,
                   A.data_ptr() + a_off * esz, lda, a_rpb, a_bs,
                   B.data_ptr(), ldb, b_rpb, b_bs,
                   out.data_ptr() + c_off * out.element_size(), ldc, c_rpb, c_bs,
                   bias.data_ptr() if bias is not None else None, 1 if bias_rows else 0,
                   act,
                   zin.data_ptr() + z_off * zin.element_size() if zin is not None else None, zout.data_ptr() if zout is not None else None,
                   ldz if ldz is not None else N, z_rpb, z_bs,
                   1 if out.dtype == torch.float32 else 0, 1 if accumulate else 0, splitk, alpha,
                   colsum.data_ptr() if colsum is not None else None)
    alg_bytes = (M * K + N * K) * esz + M * N * out.element_size() * (2 if zout is not None else 1) + (M * N * esz if zin is not None else 0)
    return d, out, alg_bytes


def gemm(layout, A, B, M, N, K, **kw):
    """C[M,N] = epi(alpha * A.B^T + bias).  A/B are CUDA tensors of the activation dtype; the logical
    shapes and leading dimensions are given explicitly so strided views (conv-as-GEMM) need no copy."""
    d, out, alg_bytes = _gemm_desc(layout, A, B, M, N, K, **kw)
    if GEMM_PROFILE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        L.check(L.lib().a8_gemm(C.byref(d), L.stream()), "a8_gemm")
        ev1.record()
        GEMM_PROFILE.append((ev0, ev1, 2.0 * M * N * K, layout, alg_bytes, (M, N, K, kw.get("splitk", 1))))
    else:
        L.check(L.lib().a8_gemm(C.byref(d), L.stream()), "a8_gemm")
    return out


def encoder_layer_fwd(x, keymask, B, T, H, p, eps, seeds, wqkv, bqkv, wo, bo, w1, b1, w2, b2, g1, be1, g2, be2, aux_stream=None):
    """a8_encoder_layer_fwd: the whole post-LN encoder layer forward in one C call.  Returns the tensors backward needs:
    (x2, qkv, o, lse, ao, x1, mean2, rstd2, z1, h, f, mean1, rstd1)."""
    M, d = x.shape
    dff = w1.shape[0]
    dev, dt = x.device, x.dtype
    L.require_cuda(x, keymask, wqkv, wo, w1, w2, bqkv, bo, b1, b2, g1, be1, g2, be2)
    qkv = torch.empty((M, 3 * d), device=dev, dtype=dt)
    o, ao, x1, f, x2 = torch.empty((5, M, d), device=dev, dtype=dt).unbind(0)          # five contiguous [M, d] slabs, one allocation
    z1, h = torch.empty((2, M, dff), device=dev, dtype=dt).unbind(0)
    lse = torch.empty((B, H, T), device=dev, dtype=torch.float32)
    stats = torch.empty((4, M), device=dev, dtype=torch.float32)
    mean2, rstd2, mean1, rstd1 = stats.unbind(0)
    sp, fsz = stats.data_ptr(), 4 * M

    def desc(b0, nb):
        """descriptor of utterances [b0, b0 + nb): every buffer advanced to their rows"""
        r0, es = b0 * T, x.element_size()
        return L.EncoderLayerDesc(L.ENC_DESC_SIZE, L.dt(x), nb, T, H, d, dff, p, eps, _u32(seeds[0]), _u32(seeds[1]), _u32(seeds[2]), b0,
                                  x.data_ptr() + r0 * d * es, keymask.data_ptr() + b0 * T if keymask is not None else None,
                                  wqkv.data_ptr(), wo.data_ptr(), w1.data_ptr(), w2.data_ptr(),
                                  bqkv.data_ptr(), bo.data_ptr(), b1.data_ptr(), b2.data_ptr(), g1.data_ptr(), be1.data_ptr(), g2.data_ptr(), be2.data_ptr(),
                                  qkv.data_ptr() + r0 * 3 * d * es, o.data_ptr() + r0 * d * es, ao.data_ptr() + r0 * d * es, x1.data_ptr() + r0 * d * es,
                                  z1.data_ptr() + r0 * dff * es, h.data_ptr() + r0 * dff * es, f.data_ptr() + r0 * d * es, x2.data_ptr() + r0 * d * es,
                                  lse.data_ptr() + 4 * b0 * H * T, sp + 4 * r0, sp + fsz + 4 * r0, sp + 2 * fsz + 4 * r0, sp + 3 * fsz + 4 * r0)
    if aux_stream is not None and B >= 2:
        # Two halves of the batch as two launch sequences on two streams (rows are independent through the whole layer): each half's
        # under-filled launches (294-tile products, latency-bound LayerNorms, the tails of every grid) run beside the other half's.
        # The caller orders the two streams at the ends of the encoder stack; inside it half A lives on the current stream, half B on aux.
        nb = B // 2
        ea, eb = desc(0, nb), desc(nb, B - nb)
        for t_ in (x, qkv, o, z1, lse, stats) + ((keymask,) if keymask is not None else ()):     # (o / z1 stand for their slabs' storage)
            t_.record_stream(aux_stream)          # allocated on the current stream, used on aux too: no reuse before aux is done with them
        L.check(L.lib().a8_encoder_layer_fwd(C.byref(ea), L.stream()), "a8_encoder_layer_fwd")
        L.check(L.lib().a8_encoder_layer_fwd(C.byref(eb), aux_stream.cuda_stream), "a8_encoder_layer_fwd")
    else:
        e = desc(0, B)
        L.check(L.lib().a8_encoder_layer_fwd(C.byref(e), L.stream()), "a8_encoder_layer_fwd")
    return x2, qkv, o, lse, ao, x1, mean2, rstd2, z1, h, f, mean1, rstd1


def encoder_layer_bwd(dx2, x, keymask, qkv, o, lse, ao, x1, mean2, rstd2, z1, f, mean1, rstd1, g1, g2, w2_t, w1_t, wo_t,
                      dg1, dbe1, dg2, dbe2, db2, db1, dbo, B, T, H, p, seeds, aux_stream=None, ln_partials=False):
    """a8_encoder_layer_bwd: LN1 bwd, FFN data gradients, LN2 bwd, output-projection data gradient and attention backward in one C call
    (two calls, utterances [0, B/2) on the current stream and [B/2, B) on aux_stream, when one is given: the parameter gradients the
    chain produces are atomic column sums, the rest is row-local).
    ln_partials: the two LayerNorm backward launches write per-block partial column sums instead of adding atomically into dg / dbe /
    db2 / dbo; the sixth return value is then the workspace f32 [2 LayerNorms][partial rows][3][d] for colsum_partials_reduce (else None).
    Returns (df, dz1, dao, dqkv, dx_res, partials): the weight-gradient operands and the residual-branch gradient of the layer input."""
    M, d = x.shape
    dff = z1.shape[1]
    dev, dt = x.device, x.dtype
    L.require_cuda(dx2, x, keymask, qkv, o, lse, ao, x1, z1, f, w2_t, w1_t, wo_t, dg1, dbe1, dg2, dbe2, db2, db1, dbo)
    n_d = 6 if p > 0 else 4
    slab = torch.empty((n_d, M, d), device=dev, dtype=dt)
    slabs = slab.unbind(0)
    df, dx1_ffn, dao, dout_o = slabs[:4]
    dx1_res, dx_res = (slabs[4], slabs[5]) if p > 0 else (df, dao)
    dz1 = torch.empty((M, dff), device=dev, dtype=dt)
    dqkv = torch.empty((M, 3 * d), device=dev, dtype=dt)
    delta = torch.empty((B, H, T), device=dev, dtype=torch.float32)
    es = x.element_size()
    split = aux_stream is not None and B >= 2
    halves = [(0, B // 2), (B // 2, B - B // 2)] if split else [(0, B)]
    ws, prow0 = None, [0] * len(halves)
    if ln_partials:
        nrows = [L.lib().a8_layernorm_bwd_partial_rows(nb * T) for _, nb in halves]
        prow0 = [sum(nrows[:i]) for i in range(len(halves))]
        ws = torch.empty((2, sum(nrows), 3, d), device=dev, dtype=torch.float32)

    def desc(i):
        b0, nb = halves[i]
        r0 = b0 * T
        rd, rf = r0 * d * es, r0 * dff * es
        wp = [ws[k].data_ptr() + prow0[i] * 3 * d * 4 for k in (0, 1)] if ws is not None else [None, None]
        return L.EncoderLayerBwdDesc(L.ENC_BWD_DESC_SIZE, L.dt(x), nb, T, H, d, dff, p, _u32(seeds[0]), _u32(seeds[1]), _u32(seeds[2]), b0,
                                     dx2.data_ptr() + rd, x.data_ptr() + rd, qkv.data_ptr() + 3 * rd, o.data_ptr() + rd, ao.data_ptr() + rd,
                                     x1.data_ptr() + rd, z1.data_ptr() + rf, f.data_ptr() + rd,
                                     keymask.data_ptr() + r0 if keymask is not None else None,
                                     lse.data_ptr() + 4 * b0 * H * T, mean2.data_ptr() + 4 * r0, rstd2.data_ptr() + 4 * r0,
                                     mean1.data_ptr() + 4 * r0, rstd1.data_ptr() + 4 * r0, g1.data_ptr(), g2.data_ptr(),
                                     w2_t.data_ptr(), w1_t.data_ptr(), wo_t.data_ptr(),
                                     dg1.data_ptr(), dbe1.data_ptr(), dg2.data_ptr(), dbe2.data_ptr(), db2.data_ptr(), db1.data_ptr(), dbo.data_ptr(),
                                     df.data_ptr() + rd, dx1_res.data_ptr() + rd, dz1.data_ptr() + rf, dx1_ffn.data_ptr() + rd, dao.data_ptr() + rd,
                                     dx_res.data_ptr() + rd, dout_o.data_ptr() + rd, dqkv.data_ptr() + 3 * rd, delta.data_ptr() + 4 * b0 * H * T,
                                     wp[0], wp[1])
    if split:
        ea, eb = desc(0), desc(1)
        for t_ in (dx2, slab, dz1, dqkv, delta) + ((ws,) if ws is not None else ()):
            t_.record_stream(aux_stream)          # (the saved forward tensors carry aux in their stream set since the forward)
        L.check(L.lib().a8_encoder_layer_bwd(C.byref(ea), L.stream()), "a8_encoder_layer_bwd")
        L.check(L.lib().a8_encoder_layer_bwd(C.byref(eb), aux_stream.cuda_stream), "a8_encoder_layer_bwd")
    else:
        e = desc(0)
        L.check(L.lib().a8_encoder_layer_bwd(C.byref(e), L.stream()), "a8_encoder_layer_bwd")
    return df, dz1, dao, dqkv, dx_res, ws


def encoder_layer_bwd_full(dx2, x, keymask, qkv, o, lse, ao, x1, mean2, rstd2, z1, h, f, mean1, rstd1, g1, g2, w2_t, w1_t, wo_t, wqkv_t,
                           dg1, dbe1, dg2, dbe2, db2, db1, dbo, dw2, dw1, dwo, dwqkv, acc, dbqkv, B, T, H, p, seeds, aux_stream, side_stream):
    """a8_encoder_layer_bwd_full: the whole backward of one encoder layer in ONE C call -- the data-gradient chains (two utterance ranges on
    the current stream and aux_stream when one is given), the grouped weight gradients + LayerNorm / bias reductions on side_stream behind
    both, the QKV data gradient per range.  Same launches, same order per stream as encoder_layer_bwd + the Python that followed it.
    acc: four flags (accumulate into / store over the weight-gradient targets).  Returns (dx, temporaries): the caller keeps the
    temporaries alive for the streams that still read them (record_stream) -- they are the operands of the side stream's launches."""
    M, d = x.shape
    dff = z1.shape[1]
    dev, dt = x.device, x.dtype
    n_d = 7 if p > 0 else 5
    slab = torch.empty((n_d, M, d), device=dev, dtype=dt)
    slabs = slab.unbind(0)
    df, dx1_ffn, dao, dout_o, dx = slabs[:5]
    dx1_res, dx_res = (slabs[5], slabs[6]) if p > 0 else (df, dao)
    dz1 = torch.empty((M, dff), device=dev, dtype=dt)
    dqkv = torch.empty((M, 3 * d), device=dev, dtype=dt)
    delta = torch.empty((B, H, T), device=dev, dtype=torch.float32)
    es = x.element_size()
    split = aux_stream is not None and B >= 2
    halves = [(0, B // 2), (B // 2, B - B // 2)] if split else [(0, B)]
    nrows = [L.lib().a8_layernorm_bwd_partial_rows(nb * T) for _, nb in halves]
    ws = torch.empty((2, sum(nrows), 3, d), device=dev, dtype=torch.float32)
    fd = L.EncoderLayerBwdFullDesc()
    fd.size, fd.nhalves, fd.M = L.ENC_BWD_FULL_DESC_SIZE, len(halves), M
    wsp = ws.data_ptr()
    prow = 0
    for i, (b0, nb) in enumerate(halves):
        r0 = b0 * T
        rd, rf = r0 * d * es, r0 * dff * es
        wp0, wp1 = wsp + prow * 3 * d * 4, wsp + (sum(nrows) + prow) * 3 * d * 4
        prow += nrows[i]
        fd.half[i] = L.EncoderLayerBwdDesc(
            L.ENC_BWD_DESC_SIZE, L.dt(x), nb, T, H, d, dff, p, _u32(seeds[0]), _u32(seeds[1]), _u32(seeds[2]), b0,
            dx2.data_ptr() + rd, x.data_ptr() + rd, qkv.data_ptr() + 3 * rd, o.data_ptr() + rd, ao.data_ptr() + rd,
            x1.data_ptr() + rd, z1.data_ptr() + rf, f.data_ptr() + rd,
            keymask.data_ptr() + r0 if keymask is not None else None,
            lse.data_ptr() + 4 * b0 * H * T, mean2.data_ptr() + 4 * r0, rstd2.data_ptr() + 4 * r0,
            mean1.data_ptr() + 4 * r0, rstd1.data_ptr() + 4 * r0, g1.data_ptr(), g2.data_ptr(),
            w2_t.data_ptr(), w1_t.data_ptr(), wo_t.data_ptr(),
            dg1.data_ptr(), dbe1.data_ptr(), dg2.data_ptr(), dbe2.data_ptr(), db2.data_ptr(), db1.data_ptr(), dbo.data_ptr(),
            df.data_ptr() + rd, dx1_res.data_ptr() + rd, dz1.data_ptr() + rf, dx1_ffn.data_ptr() + rd, dao.data_ptr() + rd,
            dx_res.data_ptr() + rd, dout_o.data_ptr() + rd, dqkv.data_ptr() + 3 * rd, delta.data_ptr() + 4 * b0 * H * T, wp0, wp1)
    fd.h, fd.wqkv_t, fd.dx = h.data_ptr(), wqkv_t.data_ptr(), dx.data_ptr()
    fd.dw2, fd.dw1, fd.dwo, fd.dwqkv = dw2.data_ptr(), dw1.data_ptr(), dwo.data_ptr(), dwqkv.data_ptr()
    fd.acc_w2, fd.acc_w1, fd.acc_wo, fd.acc_wqkv = (1 if a else 0 for a in acc)
    fd.dbqkv = dbqkv.data_ptr()
    fd.ln1_ws, fd.ln2_ws, fd.ln_rows = wsp, wsp + sum(nrows) * 3 * d * 4, sum(nrows)
    temps = (slab, dz1, dqkv, delta, ws)
    for t_ in temps + (dx2,):
        if split:
            t_.record_stream(aux_stream)
        t_.record_stream(side_stream)
    L.check(L.lib().a8_encoder_layer_bwd_full(C.byref(fd), L.stream(), aux_stream.cuda_stream if split else None, side_stream.cuda_stream),
            "a8_encoder_layer_bwd_full")
    return dx, temps


def colsum_partials_reduce(ws, out0, out1, out2):
    """out_k += column sums of ws[:, k, :] (ws f32 [rows, 3, cols]: the partial rows of a LayerNorm backward launch); on the current stream."""
    rows, three, cols = ws.shape
    assert three == 3 and ws.is_contiguous() and ws.dtype == torch.float32
    L.check(L.lib().a8_colsum_partials_reduce(ws.data_ptr(), rows, cols, L.ptr(out0), L.ptr(out1), L.ptr(out2), L.stream()), "a8_colsum_partials_reduce")


def gemm_grouped(layout, problems):
    """Several independent GEMMs of one layout and dtype in one launch.  problems: list of (A, B, M, N, K, kwargs)."""
    arr = (L.GemmDesc * len(problems))()
    flops, nbytes, outs = 0.0, 0, []
    for i, (A, B, M, N, K, kw) in enumerate(problems):
        arr[i], out, ab = _gemm_desc(layout, A, B, M, N, K, **kw)
        outs.append(out)
        flops += 2.0 * M * N * K
        nbytes += ab
    if GEMM_PROFILE is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        L.check(L.lib().a8_gemm_grouped(arr, len(problems), L.stream()), "a8_gemm_grouped")
        ev1.record()
        GEMM_PROFILE.append((ev0, ev1, flops, layout, nbytes, (-len(problems), problems[0][3], problems[0][4], 1)))
    else:
        L.check(L.lib().a8_gemm_grouped(arr, len(problems), L.stream()), "a8_gemm_grouped")
    return outs


# ----------------------------------------------------------------------------- helpers
def _u32(x):
    return int(x) & 0xFFFFFFFF


def _f(x):
    return float(x)


def conv_out_len(L, k, s):
    return (L - k) // s + 1


# ----------------------------------------------------------------------------- conv layer 0
def conv0_gn_gelu_fwd(x, w, gn_w, gn_b, out_dtype, eps=1e-5, out=None, stats=None):
    """x f32 [B,L]; w f32 [512,1,10] -> (out [B,L0,512] channels-last, stats f32 [B,512,2]); out / stats may be preallocated."""
    L.require_cuda(x, w, gn_w, gn_b, out, stats)
    B, Ls = x.shape
    Cn = w.shape[0]
    L0 = conv_out_len(Ls, 10, 5)
    if out is None:
        out = torch.empty((B, L0, Cn), device=x.device, dtype=out_dtype)
    if stats is None:
        stats = torch.empty((B, Cn, 2), device=x.device, dtype=torch.float32)
    if out.dtype != out_dtype or tuple(out.shape) != (B, L0, Cn) or tuple(stats.shape) != (B, Cn, 2) or not (out.is_contiguous() and stats.is_contiguous()):
        raise RuntimeError("a8hip conv0_gn_gelu_fwd: bad preallocated out / stats")
    part = torch.empty((B * ((L0 + 511) // 512) * 65,), device=x.device, dtype=torch.float64)      # A8_CONV0_FWD_SCRATCH_FLOATS
    L.check(L.lib().a8_conv0_gn_gelu_fwd(L.dt(out), L.ptr(_c(x)), L.ptr(_c(w)), L.ptr(gn_w), L.ptr(gn_b), L.ptr(out), L.ptr(stats),
                                         L.ptr(part), B, Ls, Cn, _f(eps), L.stream()), "a8_conv0_gn_gelu_fwd")
    return out, stats


def conv0_gn_gelu_bwd(x, w, gn_w, gn_b, stats, dout, dw, dgn_w, dgn_b):
    """Accumulates (+=) into dw f32 [512,1,10], dgn_w, dgn_b f32 [512]."""
    L.require_cuda(x, dout, dw)
    B, Ls = x.shape
    Cn = w.shape[0]
    red = torch.empty((B * (2 * 65 + 512 * 12),), device=x.device, dtype=torch.float32)     # A8_CONV0_BWD_SCRATCH_FLOATS(B)
    L.check(L.lib().a8_conv0_gn_gelu_bwd(L.dt(dout), L.ptr(_c(x)), L.ptr(_c(w)), L.ptr(gn_w), L.ptr(gn_b), L.ptr(stats), L.ptr(_c(dout)),
                                         L.ptr(dw), L.ptr(dgn_w), L.ptr(dgn_b), L.ptr(red), B, Ls, Cn, L.stream()), "a8_conv0_gn_gelu_bwd")


def col2im_gelu_grad(dcol, z, B, Lin, Lout, Cn, k, stride):
    dx = torch.empty((B, Lin, Cn), device=dcol.device, dtype=dcol.dtype)
    L.check(L.lib().a8_col2im_gelu_grad(L.dt(dcol), L.ptr(_c(dcol)), L.ptr(z), L.ptr(dx), B, Lin, Lout, Cn, k, stride, L.stream()),
            "a8_col2im_gelu_grad")
    return dx


# ----------------------------------------------------------------------------- LayerNorm
def layernorm_fwd(x, res, gamma, beta, eps, p=0.0, seed=0, rng_stream=0):
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    y = torch.empty_like(x)
    mean = torch.empty((rows,), device=x.device, dtype=torch.float32)
    rstd = torch.empty((rows,), device=x.device, dtype=torch.float32)
    L.check(L.lib().a8_layernorm_fwd(L.dt(x), L.ptr(_c(x)), L.ptr(_c(res)), L.ptr(gamma), L.ptr(beta), L.ptr(y), L.ptr(mean), L.ptr(rstd),
                                     rows, cols, _f(eps), _f(p), _u32(seed), _u32(rng_stream), L.stream()), "a8_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, dy2, x, res, gamma, mean, rstd, dgamma, dbeta, p=0.0, seed=0, rng_stream=0, want_dx=True, want_dres=True, dbias=None):
    """dbias (optional f32 [cols]) += column sums of dx: the bias gradient of the Linear that produced x."""
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    dx = torch.empty_like(x) if want_dx else None
    dres = torch.empty_like(x) if want_dres else None
    L.check(L.lib().a8_layernorm_bwd(L.dt(x), L.ptr(_c(dy)), L.ptr(_c(dy2)), L.ptr(_c(x)), L.ptr(_c(res)), L.ptr(gamma), L.ptr(mean), L.ptr(rstd),
                                     L.ptr(dx), L.ptr(dres), L.ptr(dgamma), L.ptr(dbeta), L.ptr(dbias), rows, cols, _f(p), _u32(seed), _u32(rng_stream),
                                     L.stream()), "a8_layernorm_bwd")
    return dx, dres


# ----------------------------------------------------------------------------- element-wise
def cast(src, dst_dtype):
    dst = torch.empty(src.shape, device=src.device, dtype=dst_dtype)
    L.check(L.lib().a8_cast(L.dt(src), L.dt(dst), L.ptr(_c(src)), L.ptr(dst), int(src.numel()), L.stream()), "a8_cast")
    return dst


def dropout(x, p, seed, rng_stream):
    y = torch.empty_like(x)
    L.check(L.lib().a8_dropout(L.dt(x), L.ptr(_c(x)), L.ptr(y), int(x.numel()), _f(p), _u32(seed), _u32(rng_stream), L.stream()), "a8_dropout")
    return y


def mask_fill_rows_(x2d, mask_u8, emb):
    rows, cols = x2d.shape
    L.check(L.lib().a8_mask_fill_rows(L.dt(x2d), L.ptr(_c(x2d)), L.ptr(mask_u8), L.ptr(emb), rows, cols, L.stream()), "a8_mask_fill_rows")
    return x2d


def mask_fill_rows_bwd_(dy2d, mask_u8, demb):
    rows, cols = dy2d.shape
    L.check(L.lib().a8_mask_fill_rows_bwd(L.dt(dy2d), L.ptr(_c(dy2d)), L.ptr(mask_u8), L.ptr(demb), rows, cols, L.stream()), "a8_mask_fill_rows_bwd")
    return dy2d


def mask_zero_cols_(x3d, cmask_u8):
    B, T, cols = x3d.shape
    L.check(L.lib().a8_mask_zero_cols(L.dt(x3d), L.ptr(_c(x3d)), L.ptr(cmask_u8), B, T, cols, L.stream()), "a8_mask_zero_cols")
    return x3d


def zero_rows_(x2d, keep_u8):
    rows, cols = x2d.shape
    L.check(L.lib().a8_zero_rows(L.dt(x2d), L.ptr(_c(x2d)), L.ptr(keep_u8), rows, cols, L.stream()), "a8_zero_rows")
    return x2d


def gather_rows(src2d, idx_i32):
    n, cols = idx_i32.numel(), src2d.shape[1]
    dst = torch.empty((n, cols), device=src2d.device, dtype=src2d.dtype)
    L.check(L.lib().a8_gather_rows(L.dt(src2d), L.ptr(_c(src2d)), L.ptr(idx_i32), L.ptr(dst), n, cols, L.stream()), "a8_gather_rows")
    return dst


def scatter_add_rows_(dst2d, idx_i32, src2d):
    n, cols = idx_i32.numel(), src2d.shape[1]
    L.check(L.lib().a8_scatter_add_rows(L.dt(src2d), L.ptr(_c(src2d)), L.ptr(idx_i32), L.ptr(_c(dst2d)), n, cols, L.stream()), "a8_scatter_add_rows")
    return dst2d


def add(a, b):
    y = torch.empty_like(a)
    L.check(L.lib().a8_add(L.dt(a), L.ptr(_c(a)), L.ptr(_c(b)), L.ptr(y), int(a.numel()), L.stream()), "a8_add")
    return y


def gelu_bwd_bias(dy, z, dbias):
    rows, cols = dy.numel() // dy.shape[-1], dy.shape[-1]
    dz = torch.empty_like(dy)
    L.check(L.lib().a8_gelu_bwd_bias(L.dt(dy), L.ptr(_c(dy)), L.ptr(_c(z)), L.ptr(dz), L.ptr(dbias), rows, cols, L.stream()), "a8_gelu_bwd_bias")
    return dz


# ----------------------------------------------------------------------------- positional conv
def posconv_fwd(x, w_g, bias, groups, k, pad_left, mode, want_z=False):
    """x [B,T,d]; w_g [G][dg][k][dg] (activation dtype).  mode 1: x + gelu(conv+b); 0: gelu(conv+b); -1: conv only."""
    B, T, d = x.shape
    y = torch.empty_like(x)
    z = torch.empty_like(x) if want_z else None
    L.check(L.lib().a8_posconv_fwd(L.dt(x), L.ptr(_c(x)), L.ptr(_c(w_g)), L.ptr(bias), L.ptr(y), L.ptr(z), B, T, d, groups, k, pad_left, mode,
                                   L.stream()), "a8_posconv_fwd")
    return y, z


def posconv_wgrad(x, dz, dw_g, groups, k, pad_left):
    B, T, d = x.shape
    L.check(L.lib().a8_posconv_wgrad(L.dt(x), L.ptr(_c(x)), L.ptr(_c(dz)), L.ptr(dw_g), B, T, d, groups, k, pad_left, L.stream()), "a8_posconv_wgrad")


def weightnorm_pack(v, g, groups, dtype):
    """v f32 [d, d/groups, k], g f32 [..k] -> (wf [G][co][kk][ci], wt [G][ci][kk'][co]) in `dtype`, inv_norm f32 [k]."""
    L.require_cuda(v, g)
    d, dg, k = v.shape
    wf = torch.empty((groups, dg, k, dg), device=v.device, dtype=dtype)
    wt = torch.empty((groups, dg, k, dg), device=v.device, dtype=dtype)
    inv_norm = torch.empty((k,), device=v.device, dtype=torch.float32)
    scratch = torch.empty((d * k,), device=v.device, dtype=torch.float32)
    L.check(L.lib().a8_weightnorm_pack(L.dt(wf), L.ptr(_c(v)), L.ptr(_c(g)), L.ptr(wf), L.ptr(wt), L.ptr(inv_norm), L.ptr(scratch), d, groups, k,
                                       L.stream()), "a8_weightnorm_pack")
    return wf, wt, inv_norm


def weightnorm_bwd_(dwg, v, g, inv_norm, dv, dgain, groups):
    """dv f32 [d, d/groups, k] and dgain f32 [k] += the weight-norm gradients of dwg f32 [G][co][kk][ci]."""
    d, dg, k = v.shape
    scratch = torch.empty((d * k + k,), device=v.device, dtype=torch.float32)
    L.check(L.lib().a8_weightnorm_bwd(L.ptr(_c(dwg)), L.ptr(_c(v)), L.ptr(_c(g)), L.ptr(inv_norm), L.ptr(dv), L.ptr(dgain), L.ptr(scratch), d, groups, k,
                                      L.stream()), "a8_weightnorm_bwd")


# ----------------------------------------------------------------------------- attention
def attn_fwd(q, k, v, B, T, H, ld_qkv, scale, keymask=None, p=0.0, seed=0):
    """q,k,v: [B*T, ld_qkv] views (head h at columns h*64..); returns (o [B*T, H*64], lse f32 [B,H,T])."""
    o = torch.empty((B * T, H * 64), device=q.device, dtype=q.dtype)
    lse = torch.empty((B, H, T), device=q.device, dtype=torch.float32)
    L.check(L.lib().a8_attn_fwd(L.dt(q), L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(o), L.ptr(lse), L.ptr(keymask), B, T, H, int(ld_qkv),
                                int(H * 64), _f(scale), _f(p), _u32(seed), L.stream()), "a8_attn_fwd")
    return o, lse


def attn_bwd(q, k, v, o, dout, lse, B, T, H, ld_qkv, scale, keymask=None, p=0.0, seed=0, dqkv=None, dbias=None):
    """Returns dq, dk, dv as column slices of one [B*T, 3*H*64] buffer (so the q/k/v weight-gradient GEMMs read them in place).
    dbias (optional f32 [3*H*64]) += their column sums: the q/k/v bias gradients, accumulated by the backward kernels."""
    d = H * 64
    if dqkv is None:
        dqkv = torch.empty((B * T, 3 * d), device=q.device, dtype=q.dtype)
    delta = torch.empty((B, H, T), device=q.device, dtype=torch.float32)
    dq, dk, dv = dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:]
    L.check(L.lib().a8_attn_bwd(L.dt(q), L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(o), L.ptr(_c(dout)), L.ptr(lse), L.ptr(keymask),
                                L.ptr(dq), L.ptr(dk), L.ptr(dv), L.ptr(delta), L.ptr(dbias), B, T, H, int(ld_qkv), int(d), int(3 * d),
                                _f(scale), _f(p), _u32(seed), L.stream()), "a8_attn_bwd")
    return dq, dk, dv, dqkv


# ----------------------------------------------------------------------------- VQ
def zeros(shape, dtype, device):
    """torch.zeros without an ATen fill kernel: the clear is a memset on the current stream (a8_zero)."""
    t = torch.empty(shape, device=device, dtype=dtype)
    L.check(L.lib().a8_zero(L.ptr(t), t.numel() * t.element_size(), L.stream()), "a8_zero")
    return t


def zero_(t):
    """t.zero_() as a memset on the current stream (device tensors; contiguous)."""
    if not t.is_cuda:
        return t.zero_()
    L.check(L.lib().a8_zero(L.ptr(t), t.numel() * t.element_size(), L.stream()), "a8_zero")
    return t


def conv_wgrad_fold_(dw2, grad, co, cin, k):
    """grad[co][cin][k] += dw2[co][k][cin]; dw2 is zero again afterwards."""
    L.require_cuda(dw2, grad)
    L.check(L.lib().a8_conv_wgrad_fold(L.ptr(dw2), L.ptr(grad), co, cin, k, L.stream()), "a8_conv_wgrad_fold")
    return grad


def vq_select(logits, gumbel, tau, gumbel_gen=None, seed=0):
    """gumbel: the noise to add (None: none); gumbel_gen: an output buffer -- the kernel draws the noise itself and stores it there."""
    R, V = logits.shape
    codes = torch.empty((R,), device=logits.device, dtype=torch.int32)
    avg = torch.empty((V,), device=logits.device, dtype=torch.float32)
    scratch = torch.empty((((R + 15) // 16) * V,), device=logits.device, dtype=torch.float32)        # A8_VQ_SELECT_SCRATCH_FLOATS
    L.check(L.lib().a8_vq_select(L.ptr(_c(logits)), L.ptr(_c(gumbel)), L.ptr(gumbel_gen), _u32(seed), _f(tau), L.ptr(codes), L.ptr(avg),
                                 L.ptr(scratch), R, V, L.stream()), "a8_vq_select")
    return codes, avg


def vq_gather(vars2d, codes, N, G, V, D, dtype):
    q = torch.empty((N, G * D), device=vars2d.device, dtype=dtype)
    L.check(L.lib().a8_vq_gather(L.dt(q), L.ptr(_c(vars2d)), L.ptr(codes), L.ptr(q), N, G, V, D, L.stream()), "a8_vq_gather")
    return q


def vq_ppl(avg):
    V = avg.numel()
    ppl = torch.empty((1,), device=avg.device, dtype=torch.float32)
    dppl = torch.empty((V,), device=avg.device, dtype=torch.float32)
    L.check(L.lib().a8_vq_ppl(L.ptr(avg), L.ptr(ppl), L.ptr(dppl), V, L.stream()), "a8_vq_ppl")
    return ppl, dppl


def vq_bwd(logits, gumbel, tau, codes, vars2d, dq, eproj, dppl, gppl, dvars, N, G, V, D, training):
    dlogits = torch.empty_like(logits)
    L.check(L.lib().a8_vq_bwd(L.dt(dq), L.ptr(logits), L.ptr(gumbel), _f(tau), L.ptr(codes), L.ptr(_c(vars2d)), L.ptr(_c(dq)), L.ptr(eproj),
                              L.ptr(dppl), L.ptr(gppl), L.ptr(dlogits), L.ptr(dvars), N, G, V, D, 1 if training else 0, L.stream()), "a8_vq_bwd")
    return dlogits


# ----------------------------------------------------------------------------- contrastive
def contrastive_fwd(ctx, lat, neg_idx):
    """-> (logits f32 [rows, N+1], loss_sum f32 [1], inv_norms f32 [2, rows] for contrastive_bwd)."""
    rows, Cn = ctx.shape
    N = neg_idx.shape[1]
    logits = torch.empty((rows, N + 1), device=ctx.device, dtype=torch.float32)
    loss_sum = zeros((1,), torch.float32, ctx.device)
    inv_norms = torch.empty((2, rows), device=ctx.device, dtype=torch.float32)
    L.check(L.lib().a8_contrastive_fwd(L.dt(ctx), L.ptr(_c(ctx)), L.ptr(_c(lat)), L.ptr(_c(neg_idx)), L.ptr(logits), L.ptr(loss_sum),
                                       L.ptr(inv_norms), rows, N, Cn, L.stream()), "a8_contrastive_fwd")
    return logits, loss_sum, inv_norms


def contrastive_bwd(ctx, lat, neg_idx, logits, gscale, rows_per_group=0, inv_norms=None):
    """rows_per_group: frames per utterance when every frame's negatives come from its own utterance (0: no assumption);
    inv_norms: the forward's inverse row norms (None: recomputed)."""
    rows, Cn = ctx.shape
    N = neg_idx.shape[1]
    dctx = torch.empty((rows, Cn), device=ctx.device, dtype=torch.float32)
    dlat = zeros((rows, Cn), torch.float32, ctx.device)
    scratch = torch.empty((rows * (2 + 2 * (N + 1) + 1 + int(rows_per_group) + 7 + 768) + 32,), device=ctx.device, dtype=torch.float32)    # A8_CONTRASTIVE_BWD_SCRATCH_FLOATS
    L.check(L.lib().a8_contrastive_bwd(L.dt(ctx), L.ptr(_c(ctx)), L.ptr(_c(lat)), L.ptr(_c(neg_idx)), L.ptr(logits), L.ptr(gscale), L.ptr(inv_norms),
                                       L.ptr(dctx), L.ptr(dlat), L.ptr(scratch), rows, N, Cn, int(rows_per_group), L.stream()), "a8_contrastive_bwd")
    return dctx, dlat


def loss_combine(xe_sum, ppl, rows, n_vars, w_xe, w_div):
    L.require_cuda(xe_sum, ppl)
    if xe_sum.dtype != torch.float32 or ppl.dtype != torch.float32 or xe_sum.numel() != 1 or ppl.numel() != 1:
        raise RuntimeError("a8hip loss_combine: xe_sum and ppl must be one-element f32 tensors")
    loss = torch.empty((1,), device=xe_sum.device, dtype=torch.float32)
    L.check(L.lib().a8_loss_combine(L.ptr(_c(xe_sum)), L.ptr(_c(ppl)), rows, n_vars, w_xe, w_div, L.ptr(loss), L.stream()), "a8_loss_combine")
    return loss


def loss_combine_bwd(dloss, rows, n_vars, w_xe, w_div):
    out = torch.empty((2,), device=dloss.device, dtype=torch.float32)
    L.check(L.lib().a8_loss_combine_bwd(L.ptr(dloss), rows, n_vars, w_xe, w_div, L.ptr(out), L.ptr(out[1:]), L.stream()), "a8_loss_combine_bwd")
    return out[:1], out[1:]


# ----------------------------------------------------------------------------- CTC + log-softmax
def ctc_greedy_decode(lprobs, in_len_i32, blank=0):
    """lprobs [B,T,V] (any strides on B and T, V contiguous) -> (tokens i32 [B,T], lengths i32 [B]) on the device."""
    L.require_cuda(lprobs, in_len_i32)
    B, T, V = lprobs.shape
    if lprobs.stride(2) != 1:
        lprobs = lprobs.contiguous()
    tokens = torch.empty((B, T), device=lprobs.device, dtype=torch.int32)
    lens = torch.empty((B,), device=lprobs.device, dtype=torch.int32)
    L.check(L.lib().a8_ctc_greedy_decode(L.dt(lprobs), L.ptr(lprobs), L.ptr(in_len_i32), L.ptr(tokens), L.ptr(lens), B, T, V,
                                         int(lprobs.stride(0)), int(lprobs.stride(1)), int(blank), L.stream()), "a8_ctc_greedy_decode")
    return tokens, lens



def ctc_pack_targets(targets_i64, tlen_i32, pad, eos):
    """targets [B,S] int64 (device) -> (flat i32 labels, row b at offsets[b]; offsets i32 [B+1]; status i32 [2] = (rows whose
    PAD/EOS count disagrees with target_lengths, max(target_lengths)) for a deferred check)."""
    L.require_cuda(targets_i64, tlen_i32)
    B, S = targets_i64.shape
    flat = zeros((max(1, B * S),), torch.int32, targets_i64.device)          # (labels of a short row stay defined)
    off = torch.empty((B + 1,), device=targets_i64.device, dtype=torch.int32)
    status = torch.empty((2,), device=targets_i64.device, dtype=torch.int32)
    L.check(L.lib().a8_ctc_pack_targets(L.ptr(_c(targets_i64)), L.ptr(tlen_i32), L.ptr(flat), L.ptr(off), L.ptr(status), B, S, int(pad), int(eos),
                                        L.stream()), "a8_ctc_pack_targets")
    return flat, off, status


def ctc_fwd_bwd(lp, targets_flat_i32, tgt_off_i32, in_len_i32, Smax, blank=0, zero_infinity=True):
    T, B, V = lp.shape
    nll = torch.empty((B,), device=lp.device, dtype=torch.float32)
    grad = torch.empty_like(lp)
    # A8_CTC_SCRATCH_FLOATS(T, B, Smax): normalised log2 alpha | beta, [B][T][64 ceil((2 Smax + 1) / 64)] each, + a flag per utterance
    alpha = torch.empty((2 * B * T * 64 * ((2 * Smax + 1 + 63) // 64) + B,), device=lp.device, dtype=torch.float32)
    L.check(L.lib().a8_ctc_fwd_bwd(L.ptr(_c(lp)), L.ptr(targets_flat_i32), L.ptr(tgt_off_i32), L.ptr(in_len_i32), L.ptr(nll), L.ptr(grad),
                                   L.ptr(alpha), T, B, V, Smax, blank, 1 if zero_infinity else 0, L.stream()), "a8_ctc_fwd_bwd")
    return nll, grad


def log_softmax_fwd(x2d, cols=None):
    """x2d [rows, ldx]: log_softmax over the first `cols` columns (the rest is GEMM padding) -> f32 [rows, cols]."""
    rows, ldx = x2d.shape
    cols = ldx if cols is None else cols
    y = torch.empty((rows, cols), device=x2d.device, dtype=torch.float32)
    L.check(L.lib().a8_log_softmax_fwd(L.ptr(_c(x2d)), L.dt(x2d), L.ptr(y), rows, cols, ldx, L.stream()), "a8_log_softmax_fwd")
    return y


def log_softmax_bwd(y, dy, dtype, ldx=None):
    rows, cols = y.shape
    ldx = cols if ldx is None else ldx
    dx = torch.empty((rows, ldx), device=y.device, dtype=dtype)
    L.check(L.lib().a8_log_softmax_bwd(L.ptr(y), L.ptr(_c(dy)), L.ptr(dx), L.dt(dx), rows, cols, ldx, L.stream()), "a8_log_softmax_bwd")
    return dx


# ----------------------------------------------------------------------------- optimizer
def sumsq_(x, out):
    L.check(L.lib().a8_sumsq(L.ptr(x), int(x.numel()), L.ptr(out), L.stream()), "a8_sumsq")


def adamw_(p, g, m, v, lr, beta1, beta2, eps, wd, step, gnorm_sq=None, max_norm=0.0, grad_scale=1.0, zero_grad=False, grad_denom=None):
    """grad_denom: optional int64 device tensor; gradients are divided by its first element on the device."""
    if grad_denom is not None and (grad_denom.dtype != torch.int64 or not grad_denom.is_cuda):
        raise RuntimeError("a8hip adamw_: grad_denom must be an int64 device tensor")
    L.check(L.lib().a8_adamw(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), int(p.numel()), _f(lr), _f(beta1), _f(beta2), _f(eps), _f(wd),
                             int(step), L.ptr(gnorm_sq), _f(max_norm), _f(grad_scale), L.ptr(grad_denom), 1 if zero_grad else 0, L.stream()), "a8_adamw")


# ----------------------------------------------------------------------------- weight preparation
def prepare_weights(dtype_code, desc_array, n):
    L.check(L.lib().a8_prepare_weights(dtype_code, desc_array, n, L.stream()), "a8_prepare_weights")
