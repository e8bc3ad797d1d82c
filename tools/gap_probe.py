#!/usr/bin/env python3
"""GPU-side cost of a dependent kernel boundary: N tiny kernels queued behind a long one (host never the bottleneck)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import _lib as L, kernels as K  # noqa: E402

bf = torch.bfloat16
A = torch.randn(16384, 8192, device="cuda").to(bf)
B = torch.randn(8192, 8192, device="cuda").to(bf)
out = torch.empty(16384, 8192, device="cuda", dtype=bf)
x = torch.randn(64, 64, device="cuda").to(bf)
y = torch.randn(64, 64, device="cuda").to(bf)


def run(n):
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    for _ in range(3):
        K.gemm(L.NT, A, B, 16384, 8192, 8192, out=out)      # ~6 ms of queue: the host gets far ahead
    e1.record()
    for _ in range(n):
        K.add(x, y)
    e2.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), e1.elapsed_time(e2)


for n in (100, 400):
    run(n)
    big, small = run(n)
    print("n=%d: long kernels %.2f ms, %d tiny dependent kernels %.3f ms -> %.2f us each" % (n, big, n, small, small * 1e3 / n))
