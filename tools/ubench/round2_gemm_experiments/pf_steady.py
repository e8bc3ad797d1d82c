#!/usr/bin/env python3
"""Steady-state main-loop rate of the default and the register-pipelined GEMM kernels: long K, few tiles."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import _lib as L, kernels as K  # noqa: E402
from tools.micro_bench import timeit  # noqa: E402

bf = torch.bfloat16
for (M, N, Kd) in [(4096, 4096, 16384), (8192, 4096, 4096), (4096, 2048, 32768), (16384, 2048, 2048)]:
    A = torch.randn(M, Kd, device="cuda").to(bf)
    B = torch.randn(N, Kd, device="cuda").to(bf)
    out = torch.empty(M, N, device="cuda", dtype=bf)
    res = []
    for rnd in range(2):
        for pf in (0, 1):
            L.check(L.lib().a8_set_tuning(5, pf), "tune")
            t = timeit(lambda: K.gemm(L.NT, A, B, M, N, Kd, out=out), iters=10)
            res.append("pf=%d %.0f us %.0f TF/s" % (pf, t, 2.0 * M * N * Kd / t / 1e6))
    print(M, N, Kd, " | ".join(res))
L.check(L.lib().a8_set_tuning(5, 0), "tune")
