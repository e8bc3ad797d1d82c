#!/usr/bin/env python3
"""Yardstick, not product: what the vendor library (hipBLASLt / rocBLAS behind torch.matmul) reaches on this step's plain
GEMM shapes, next to a8_gemm on the same operands, interleaved in one process.  Tells how much of the gap to the MFMA peak is
the shapes' (nobody gets it) and how much is this kernel's.  Usage (GPU box): python tools/vendor_yardstick.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import _lib as L, kernels as K  # noqa: E402

M = 6248
SHAPES = [  # name, M, N, K   (C[M,N] = A[M,K] . B[N,K]^T, the Linear forward / the NT data gradient)
    ("qkv", M, 2304, 768),
    ("o", M, 768, 768),
    ("ffn1", M, 3072, 768),
    ("ffn2", M, 768, 3072),
    ("dqkv", M, 768, 2304),
    ("conv1-like", 8 * 24999, 512, 1536),
    ("conv2-like", 8 * 12499, 512, 1536),
    ("conv5-like", 8 * 1562, 512, 1024),
    ("4096^2x8192", 4096, 4096, 8192),
    ("8192x4096x4096", 8192, 4096, 4096),
]
WG = [  # weight gradients: C[N,K] (f32 in the product) = dY[M,N]^T . X[M,K]
    ("dW1", M, 3072, 768),
    ("dWo", M, 768, 768),
    ("dWconv1", 8 * 24999, 512, 1536),
]


def timeit(f, iters=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    dev, bf = "cuda", torch.bfloat16
    print("%-16s %10s %8s   %10s %8s" % ("shape", "vendor us", "TF/s", "a8 us", "TF/s"))
    for name, m, n, k in SHAPES:
        A, B = torch.randn(m, k, device=dev).to(bf), torch.randn(n, k, device=dev).to(bf)
        Cv = torch.empty(m, n, device=dev, dtype=bf)
        Ca = torch.empty(m, n, device=dev, dtype=bf)
        res = []
        for rnd in range(2):
            tv = timeit(lambda: torch.matmul(A, B.t(), out=Cv))
            ta = timeit(lambda: K.gemm(L.NT, A, B, m, n, k, out=Ca))
            res.append((tv, ta))
        tv, ta = min(r[0] for r in res), min(r[1] for r in res)
        fl = 2.0 * m * n * k
        print("%-16s %10.1f %8.1f   %10.1f %8.1f   maxdiff %.3g" % (name, tv, fl / tv / 1e6, ta, fl / ta / 1e6, float((Cv.float() - Ca.float()).abs().max())))
    for name, m, n, k in WG:
        dY, X = torch.randn(m, n, device=dev).to(bf), torch.randn(m, k, device=dev).to(bf)
        Cv = torch.empty(n, k, device=dev, dtype=bf)
        Ca = torch.zeros(n, k, device=dev, dtype=torch.float32)
        tv = timeit(lambda: torch.matmul(dY.t(), X, out=Cv))
        ta = timeit(lambda: K.gemm(L.TN, dY, X, n, k, m, out=Ca, splitk=max(1, 512 // ((n // 128) * (k // 128))), accumulate=True))
        fl = 2.0 * m * n * k
        print("%-16s %10.1f %8.1f   %10.1f %8.1f   (TN, a8 accumulates in f32 with split-K atomics)" % (name, tv, fl / tv / 1e6, ta, fl / ta / 1e6))


if __name__ == "__main__":
    main()
