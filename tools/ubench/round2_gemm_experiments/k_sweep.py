#!/usr/bin/env python3
"""Fixed cost vs per-k-tile cost of the NT GEMM at the encoder's M = 6248: time(K) for several K at N = 3072 and 768."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import _lib as L, kernels as K  # noqa: E402
from tools.micro_bench import timeit  # noqa: E402

bf = torch.bfloat16
M = 6248
for N in (3072, 768, 2304):
    out = torch.empty(M, N, device="cuda", dtype=bf)
    res = []
    for Kd in (64, 256, 768, 1536, 3072, 6144):
        A = torch.randn(M, Kd, device="cuda").to(bf)
        B = torch.randn(N, Kd, device="cuda").to(bf)
        t = min(timeit(lambda: K.gemm(L.NT, A, B, M, N, Kd, out=out), iters=30) for _ in range(2))
        res.append((Kd, t))
    (k0, t0), (k1, t1) = res[2], res[-1]
    slope = (t1 - t0) / (k1 - k0) * 64
    print("N=%d: " % N + "  ".join("K=%d %.1fus" % r for r in res) + "   -> %.2f us per k-tile, intercept %.1f us" % (slope, t0 - slope * k0 / 64))
