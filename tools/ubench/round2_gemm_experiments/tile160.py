#!/usr/bin/env python3
"""NT GEMMs with between one and two 128-row tiles per CU (N = 768 encoder products: 294 tiles on 256 CUs): default tiles vs
160-row tiles (a8_set_tuning key 7 = 1).  Interleaved rounds, results must be bit-identical."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import _lib as L, kernels as K  # noqa: E402
from tools.micro_bench import timeit  # noqa: E402

bf = torch.bfloat16
M = int(os.environ.get("M", "6248"))
for N, Kd in ((768, 768), (768, 2304), (768, 3072), (640, 512)):
    A = torch.randn(M, Kd, device="cuda").to(bf)
    B = torch.randn(N, Kd, device="cuda").to(bf)
    bias = torch.randn(N, device="cuda")
    outs = {}
    res = []
    for rnd in range(2):
        for thr in (0, 2, 4):
            L.check(L.lib().a8_set_tuning(7, thr), "tune")
            out = torch.empty(M, N, device="cuda", dtype=bf)
            t = timeit(lambda: K.gemm(L.NT, A, B, M, N, Kd, out=out, bias=bias), iters=30)
            outs[thr] = out
            res.append("key7=%d %.1f us (%.0f TF/s)" % (thr, t, 2.0 * M * N * Kd / t / 1e6))
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[0], outs[4])
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    print("M=%d N=%d K=%d tiles=%d: " % (M, N, Kd, tiles) + " | ".join(res), flush=True)
L.check(L.lib().a8_set_tuning(7, 0), "tune")
