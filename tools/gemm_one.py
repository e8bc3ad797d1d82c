#!/usr/bin/env python3
"""Launch a few representative bf16 GEMMs (for PMC runs)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import _lib as L, kernels as K
M = 6248
d = "cuda"
def mk(*s): return torch.randn(*s, device=d).to(torch.bfloat16)
A, B1, B2 = mk(M, 768), mk(3072, 768), mk(768, 768)
A2 = mk(200000, 1536); B3 = mk(512, 1536)
for _ in range(3):
    K.gemm(L.NT, A, B1, M, 3072, 768)          # ffn1-like, 1176 blocks
    K.gemm(L.NT, A, B2, M, 768, 768)           # o-proj-like, 294 blocks
    K.gemm(L.NT, A2, B3, 200000, 512, 1536)    # conv1-like, 6252 blocks
torch.cuda.synchronize()
