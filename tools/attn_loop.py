#!/usr/bin/env python3
"""50 forward + backward attention calls at the cfg2 shape (for rocprofv3 --kernel-trace --stats: per-kernel durations alone)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import kernels as K  # noqa: E402

B, T, H = 8, 781, 12
dm = H * 64
bf = torch.bfloat16
qkv = (torch.randn(B * T, 3 * dm, device="cuda") * 0.5).to(bf)
dout = torch.randn(B * T, dm, device="cuda").to(bf)
p = float(os.environ.get("P", "0.1"))
dq = torch.empty(B * T, 3 * dm, device="cuda", dtype=bf)
for _ in range(50):
    o, lse = K.attn_fwd(qkv[:, :dm], qkv[:, dm:2 * dm], qkv[:, 2 * dm:], B, T, H, 3 * dm, 0.125, None, p=p, seed=3)[:2]
    K.attn_bwd(qkv[:, :dm], qkv[:, dm:2 * dm], qkv[:, 2 * dm:], o, dout, lse, B, T, H, 3 * dm, 0.125, None, p=p, seed=3, dqkv=dq)
torch.cuda.synchronize()
