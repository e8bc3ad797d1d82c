#!/usr/bin/env python3
"""conv layer 0 (+GroupNorm+GELU) forward / backward at the cfg2 size: VALU form vs MFMA form (a8_set_tuning key 9), same process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import _lib as L, kernels as K  # noqa: E402
from tools.micro_bench import timeit  # noqa: E402

B, Ls = 8, 250000
x = (0.1 * torch.randn(B, Ls, generator=torch.Generator().manual_seed(0))).clamp_(-1, 1).cuda()
w = (torch.randn(512, 1, 10) * (6.0 / 10) ** 0.5).cuda()
gw, gb = (1 + 0.1 * torch.randn(512)).cuda(), (0.1 * torch.randn(512)).cuda()
outs = {}
for v in (1, 0):
    L.check(L.lib().a8_set_tuning(9, v), "tune")
    out, stats = K.conv0_gn_gelu_fwd(x, w, gw, gb, torch.bfloat16)
    outs[v] = out.float()
    dout = torch.randn_like(out)
    dw, dg, db = torch.zeros_like(w), torch.zeros(512, device="cuda"), torch.zeros(512, device="cuda")
    tf = timeit(lambda: K.conv0_gn_gelu_fwd(x, w, gw, gb, torch.bfloat16), iters=20)
    tb = timeit(lambda: K.conv0_gn_gelu_bwd(x, w, gw, gb, stats, dout, dw, dg, db), iters=20)
    nbytes = out.numel() * 2
    print("key9=%d (%s): fwd %.1f us (%.2f TB/s written), bwd %.1f us (%.2f TB/s read)" % (v, "VALU" if v else "MFMA", tf, nbytes / tf / 1e6, tb, nbytes / tb / 1e6))
d = (outs[0] - outs[1]).abs()
print("max |MFMA - VALU| = %.4g (bf16 outputs, |out| max %.3g); mismatching elements %.3g" % (d.max().item(), outs[1].abs().max().item(), (d > 0).float().mean().item()))
L.check(L.lib().a8_set_tuning(9, 0), "tune")
