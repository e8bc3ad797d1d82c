#!/usr/bin/env python3
"""Sweep split-K for the weight-gradient (TN) shapes in one process."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import _lib as L, kernels as K
M = 6248
d = "cuda"
def mk(*s): return torch.randn(*s, device=d).to(torch.bfloat16)
def t(fn, it=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / it
for name, m, n, k, sks in [("dW1", 3072, 768, M, (1, 2, 3, 4, 6)), ("dWo", 768, 768, M, (2, 3, 4, 5, 7, 10, 14)), ("dWqkv", 2304, 768, M, (1, 2, 3, 4)),
                           ("dWfinal", 256, 768, M, (4, 8, 12, 16)), ("conv1", 512, 1536, 8 * 24999, (4, 5, 8, 10, 16, 20)), ("conv4", 512, 1536, 8 * 3124, (4, 8, 10, 16))]:
    A, B = mk(k, m), mk(k, n)
    C = torch.zeros(m, n, device=d)
    tiles = ((m + 127) // 128) * ((n + 127) // 128)
    print(name, "tiles", tiles, "  ".join(f"sk{sk}({tiles * sk} blk): {t(lambda: K.gemm(L.TN, A, B, m, n, k, out=C, splitk=sk, accumulate=True)):.1f}us" for sk in sks))
