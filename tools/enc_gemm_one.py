#!/usr/bin/env python3
"""One variant of the encoder's wide NT products alone on the GPU, for counter passes (tools/pmc_wide.sh): python3 tools/enc_gemm_one.py <key>=<val> ..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import _lib as L, kernels as K  # noqa: E402

for kv in sys.argv[1:]:
    k, v = kv.split("=")
    L.check(L.lib().a8_set_tuning(int(k), int(v)), "tune")
dev, bf, M, d = "cuda", torch.bfloat16, 6248, 768
x = torch.randn(M, d, device=dev).to(bf)
wq, w1 = torch.randn(3 * d, d, device=dev).to(bf), torch.randn(4 * d, d, device=dev).to(bf)
bq, b1 = torch.randn(3 * d, device=dev), torch.randn(4 * d, device=dev)
o3, o4, z4 = torch.empty(M, 3 * d, device=dev, dtype=bf), torch.empty(M, 4 * d, device=dev, dtype=bf), torch.empty(M, 4 * d, device=dev, dtype=bf)
for _ in range(20):
    K.gemm(L.NT, x, wq, M, 3 * d, d, bias=bq, out=o3)
    K.gemm(L.NT, x, w1, M, 4 * d, d, bias=b1, act=L.ACT_GELU, zout=z4, out=o4)
torch.cuda.synchronize()
