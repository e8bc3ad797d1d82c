#!/usr/bin/env python3
"""Weight-gradient (TN) products of the step: default 128 x 128 tiles vs the 256 x 256 tiles of a8_set_tuning key 18, interleaved."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import _lib as L, kernels as K, ops  # noqa: E402
from tools.micro_bench import timeit  # noqa: E402

bf, dev = torch.bfloat16, "cuda"
M = 6248


def ab(name, fn, flops):
    t = {}
    for rnd in range(3):
        for mode in (0, 1):
            L.check(L.lib().a8_set_tuning(18, mode), "tune")
            t[mode] = min(t.get(mode, 1e9), timeit(fn, iters=10))
    print("%-36s default %7.1f us (%4.0f TF)   256x256 %7.1f us (%4.0f TF)" % (name, t[0], flops / t[0] / 1e6, t[1], flops / t[1] / 1e6), flush=True)


shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
ops_ = [(torch.randn(M, m, device=dev).to(bf), torch.randn(M, n, device=dev).to(bf), torch.zeros(m, n, device=dev)) for m, n in shapes]
ab("encoder layer, grouped (4 products)", lambda: K.gemm_grouped(L.TN, [(a, b, m, n, M, dict(out=o, accumulate=True)) for (a, b, o), (m, n) in zip(ops_, shapes)]),
   sum(2.0 * M * m * n for m, n in shapes))
for (m, n), (a, b, o) in zip(shapes, ops_):
    sk = ops._splitk(ops._tiles(m, n), M)
    ab("dW %d x %d, K = %d (split %d)" % (m, n, M, sk), lambda: K.gemm(L.TN, a, b, m, n, M, out=o, accumulate=True, splitk=sk), 2.0 * M * m * n)
for Lin, Lout, k in [(49999, 24999, 3), (24999, 12499, 3), (6249, 3124, 3)]:
    x = torch.randn(8 * Lin * 512, device=dev).to(bf)
    dz = torch.randn(8 * Lout, 512, device=dev).to(bf)
    o = torch.zeros(512, k * 512, device=dev)
    sk = ops._splitk(ops._tiles(512, k * 512), 8 * Lout)
    ab("conv wgrad Lout = %d (split %d)" % (Lout, sk), lambda: K.gemm(L.TN, dz, x, 512, k * 512, 8 * Lout, out=o, accumulate=True, splitk=sk, ldb=2 * 512, b_rpb=Lout, b_bs=Lin * 512),
       2.0 * 8 * Lout * 512 * k * 512)
L.check(L.lib().a8_set_tuning(18, 0), "tune")
