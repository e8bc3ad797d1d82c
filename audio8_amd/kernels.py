"""Thin, autograd-free Python wrappers over the C ABI (one function per entry point).

Tensors are torch CUDA tensors used only as device-memory handles; every wrapper launches on
torch's current HIP stream.  All arithmetic happens inside liba8hip.so.
"""
import ctypes as C

import torch

from . import _lib as L


def _c(t):
    if t is not None and not t.is_contiguous():
        raise RuntimeError("a8hip: expected a contiguous tensor")
    return t


def gemm(layout, A, B, M, N, K, *, out=None, out_dtype=None, lda=None, ldb=None, ldc=None, bias=None, bias_rows=False,
         act=L.ACT_NONE, zin=None, zout=None, ldz=None, accumulate=False, splitk=1, alpha=1.0,
         a_rpb=0, a_bs=0, b_rpb=0, b_bs=0, c_rpb=0, c_bs=0):
    """C[M,N] = epi(alpha * A.B^T + bias).  A/B are CUDA tensors of the activation dtype; the logical
    shapes and leading dimensions are given explicitly so strided views (conv-as-GEMM) need no copy."""
    L.require_cuda(A, B, out, bias, zin, zout)
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
    d = L.GemmDesc()
    d.dtype, d.layout, d.M, d.N, d.K = dtype, layout, M, N, K
    d.A, d.lda, d.a_rpb, d.a_bs = A.data_ptr(), lda, a_rpb, a_bs
    d.B, d.ldb, d.b_rpb, d.b_bs = B.data_ptr(), ldb, b_rpb, b_bs
    d.C, d.ldc, d.c_rpb, d.c_bs = out.data_ptr(), ldc, c_rpb, c_bs
    d.bias = bias.data_ptr() if bias is not None else None
    d.bias_rows = 1 if bias_rows else 0
    d.act = act
    d.zin = zin.data_ptr() if zin is not None else None
    d.zout = zout.data_ptr() if zout is not None else None
    d.ldz = ldz if ldz is not None else N
    d.out_f32 = 1 if out.dtype == torch.float32 else 0
    d.accumulate = 1 if accumulate else 0
    d.splitk = splitk
    d.alpha = alpha
    L.check(L.lib().a8_gemm(C.byref(d), L.stream()), "a8_gemm")
    return out
