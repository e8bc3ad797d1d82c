#!/usr/bin/env python3
"""A few launches of the conv-layer-1 forward GEMM (M = 8 x 24999 rows over the channels-last layer-0 output, N = 512,
K = 1536, overlapping rows) for a rocprofv3 --pmc pass: which is it short of, HBM bytes or in-core cycles?
Prints event timings; run under rocprofv3 to get FETCH_SIZE / WRITE_SIZE per dispatch (tools/conv_gemm_probe.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import _lib as L, kernels as K  # noqa: E402
from tools.micro_bench import timeit  # noqa: E402

bf = torch.bfloat16
B, Lin, C, k, s = 8, 49999, 512, 3, 2
Lout = (Lin - k) // s + 1
a = torch.randn(B, Lin, C, device="cuda").to(bf)
w = torch.randn(C, k * C, device="cuda").to(bf)
M = B * Lout
for gm in [int(v) for v in os.environ.get("VARIANTS", "0").split(",")]:
    L.check(L.lib().a8_set_tuning(0, gm if gm else 5), "tune")         # GEMM stage-geometry variant (0 -> the default, 5)
    y = torch.empty(M, C, device="cuda", dtype=bf)
    z = torch.empty(M, C, device="cuda", dtype=bf)
    for _ in range(2):
        timeit(lambda: K.gemm(L.NT, a, w, M, C, k * C, lda=s * C, a_rpb=Lout, a_bs=Lin * C, out=y), iters=5)
    t = timeit(lambda: K.gemm(L.NT, a, w, M, C, k * C, lda=s * C, a_rpb=Lout, a_bs=Lin * C, out=y), iters=5)
    t2 = timeit(lambda: K.gemm(L.NT, a, w, M, C, k * C, lda=s * C, a_rpb=Lout, a_bs=Lin * C, act=L.ACT_GELU, zout=z, out=y), iters=5)
    fl = 2.0 * M * C * k * C
    print("variant %d: plain %.1f us (%.0f TF/s) | gelu+zout %.1f us (%.0f TF/s); unique bytes: A %.0f MB, C %.0f MB" %
          (gm, t, fl / t / 1e6, t2, fl / t2 / 1e6, a.numel() * 2 / 1e6, M * C * 2 / 1e6), flush=True)
