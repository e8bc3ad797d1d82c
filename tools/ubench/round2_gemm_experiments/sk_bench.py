#!/usr/bin/env python3
"""Stream-K (a8_set_tuning key 16) against the default tiling on the encoder's M = 6248 products, interleaved in one process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import _lib as L, kernels as K  # noqa: E402
from tools.micro_bench import timeit  # noqa: E402

bf, dev = torch.bfloat16, "cuda"
K.STREAM_K = True
M = 6248
shapes = [("qkv", M, 2304, 768), ("o", M, 768, 768), ("ffn1", M, 3072, 768), ("ffn2", M, 768, 3072), ("dqkv", M, 768, 2304), ("large-ffn1", 3996, 4096, 1024), ("large-o", 3996, 1024, 1024)]
for name, m, n, k in shapes:
    A, B = torch.randn(m, k, device=dev).to(bf), torch.randn(n, k, device=dev).to(bf)
    out = torch.empty(m, n, device=dev, dtype=bf)
    bias = torch.randn(n, device=dev)
    res = {}
    for rnd in range(3):
        for mode in (0, 1):
            L.check(L.lib().a8_set_tuning(16, mode), "tune")
            t = timeit(lambda: K.gemm(L.NT, A, B, m, n, k, out=out, bias=bias), iters=30)
            res[mode] = min(res.get(mode, 1e9), t)
    fl = 2.0 * m * n * k
    print("%-12s default %6.1f us (%4.0f TF)   stream-K %6.1f us (%4.0f TF)" % (name, res[0], fl / res[0] / 1e6, res[1], fl / res[1] / 1e6), flush=True)
L.check(L.lib().a8_set_tuning(16, 0), "tune")
