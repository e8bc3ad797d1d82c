#!/usr/bin/env python3
"""Only the conv feature extractor forward at the cfg2 size (bf16), a few times: every kernel of this process belongs to the
conv-stack forward, so PMC passes over it give the component's HBM traffic (profiles/rNN_gemm_traffic.json: conv_extractor_fwd_hbm_bytes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import ops, wav2vec2 as W  # noqa: E402

ops.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
fx = W.ConvFeatureExtractionModel(W.CONV_FEATURES[16]).cuda()
x = (0.1 * torch.randn(8, 250000, generator=torch.Generator().manual_seed(1234))).clamp_(-1, 1).cuda()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
with torch.no_grad():
    for _ in range(n):
        y = fx(x)
torch.cuda.synchronize()
print("conv forward passes:", n, tuple(y.shape))
