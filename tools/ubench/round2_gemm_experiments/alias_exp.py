#!/usr/bin/env python3
"""Where does a short-M NT GEMM's k-tile time go?  Same launch with the A and/or B operand aliased to a single row
(lda / ldb = 0: every DMA hits the same few cache lines) against the real strides."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import _lib as L, kernels as K  # noqa: E402
from tools.micro_bench import timeit  # noqa: E402

bf = torch.bfloat16
for M, N, Kd in ((6248, 768, 3072), (6248, 3072, 768), (6248, 768, 768), (199992, 512, 1536)):
    A = torch.randn(M, Kd, device="cuda").to(bf)
    B = torch.randn(N, Kd, device="cuda").to(bf)
    out = torch.empty(M, N, device="cuda", dtype=bf)
    res = []
    for name, kw in (("real", {}), ("A aliased", dict(lda=0)), ("B aliased", dict(ldb=0)), ("both aliased", dict(lda=0, ldb=0))):
        t = min(timeit(lambda: K.gemm(L.NT, A, B, M, N, Kd, out=out, **kw), iters=20) for _ in range(2))
        res.append("%s %.1f us" % (name, t))
    print("M=%d N=%d K=%d: " % (M, N, Kd) + " | ".join(res), flush=True)
