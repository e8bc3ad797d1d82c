#!/usr/bin/env python3
"""Steady-state rate of the GEMM kernel per operand layout (NT / NN / TN) on shapes without shape effects, plus the
encoder weight-gradient shapes (TN, reduction over M = 6248) alone and as the grouped launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import _lib as L, kernels as K  # noqa: E402
from tools.micro_bench import timeit  # noqa: E402

bf = torch.bfloat16
d = "cuda"


def rate(t, M, N, Kd):
    return 2.0 * M * N * Kd / t / 1e6


for (M, N, Kd) in [(4096, 4096, 8192), (8192, 4096, 4096)]:
    A_mk = torch.randn(M, Kd, device=d).to(bf)
    A_km = torch.randn(Kd, M, device=d).to(bf)
    B_nk = torch.randn(N, Kd, device=d).to(bf)
    B_kn = torch.randn(Kd, N, device=d).to(bf)
    out = torch.empty(M, N, device=d, dtype=bf)
    outf = torch.zeros(M, N, device=d, dtype=torch.float32)
    res = []
    for rnd in range(2):
        t = timeit(lambda: K.gemm(L.NT, A_mk, B_nk, M, N, Kd, out=out), iters=10)
        res.append("NT %.0f" % rate(t, M, N, Kd))
        t = timeit(lambda: K.gemm(L.NN, A_mk, B_kn, M, N, Kd, out=out), iters=10)
        res.append("NN %.0f" % rate(t, M, N, Kd))
        t = timeit(lambda: K.gemm(L.TN, A_km, B_kn, M, N, Kd, out=out), iters=10)
        res.append("TN %.0f" % rate(t, M, N, Kd))
        t = timeit(lambda: K.gemm(L.TN, A_km, B_kn, M, N, Kd, out=outf, accumulate=True), iters=10)
        res.append("TN+acc(f32) %.0f" % rate(t, M, N, Kd))
    print(M, N, Kd, "TF/s:", " | ".join(res), flush=True)

# encoder weight gradients: dW[n_out, n_in] += dy[m, n_out]^T x[m, n_in]
m = 6248
shapes = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]
ops_ = []
for (no, ni) in shapes:
    dy = torch.randn(m, no, device=d).to(bf)
    x = torch.randn(m, ni, device=d).to(bf)
    tgt = torch.zeros(no, ni, device=d, dtype=torch.float32)
    ops_.append((dy, x, no, ni, tgt))
    for sk in (1, 2, 4):
        t = timeit(lambda: K.gemm(L.TN, dy, x, no, ni, m, out=tgt, accumulate=True, splitk=sk), iters=20)
        print("wgrad %dx%d splitk %d: %.1f us %.0f TF/s" % (no, ni, sk, t, rate(t, no, ni, m)), flush=True)
t = timeit(lambda: K.gemm_grouped(L.TN, [(dy, x, no, ni, m, dict(out=tgt, accumulate=True)) for dy, x, no, ni, tgt in ops_]), iters=20)
fl = sum(2.0 * no * ni * m for _, _, no, ni, _ in ops_)
print("grouped wgrad: %.1f us %.0f TF/s" % (t, fl / t / 1e6))
