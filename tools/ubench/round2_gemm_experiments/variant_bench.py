#!/usr/bin/env python3
"""GEMM stage-geometry variants (a8_set_tuning key 0) on the step's shapes, interleaved rounds in one process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import _lib as L, kernels as K  # noqa: E402
from tools.micro_bench import timeit  # noqa: E402

bf = torch.bfloat16
variants = [int(v) for v in os.environ.get("VARIANTS", "5,7").split(",")]
key = int(os.environ.get("KEY", "0"))          # which tuning key the values go to (0: stage-geometry variant)
shapes = [(6248, 768, 768), (6248, 768, 3072), (6248, 3072, 768), (6248, 2304, 768), (199992, 512, 1536), (99992, 512, 1536), (4096, 4096, 8192)]
for M, N, Kd in shapes:
    A = torch.randn(M, Kd, device="cuda").to(bf)
    B = torch.randn(N, Kd, device="cuda").to(bf)
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=bf)
    z = torch.empty(M, N, device="cuda", dtype=bf)
    res, ref = [], None
    for rnd in range(3):
        for v in variants:
            L.check(L.lib().a8_set_tuning(key, v), "tune")
            t = timeit(lambda: K.gemm(L.NT, A, B, M, N, Kd, out=out, bias=bias, act=L.ACT_GELU, zout=z), iters=10)
            if rnd == 2:
                res.append("v%d %.1f us (%.0f TF/s)" % (v, t, 2.0 * M * N * Kd / t / 1e6))
            if ref is None:
                ref = out.clone()
            else:
                assert torch.equal(ref, out), "variant %d differs" % v
    print("M=%d N=%d K=%d (bias+GELU+zout): " % (M, N, Kd) + " | ".join(res), flush=True)
L.check(L.lib().a8_set_tuning(key, 5 if key == 0 else 0), "tune")
