#!/usr/bin/env python3
"""posconv weight-gradient loop-split sweep (a8_set_tuning key 6)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import _lib as L, kernels as K  # noqa: E402
from tools.micro_bench import timeit  # noqa: E402

B, T, d, G, k = 8, 781, 768, 16, 128
x = torch.randn(B, T, d, device="cuda").to(torch.bfloat16)
dz = torch.randn(B, T, d, device="cuda").to(torch.bfloat16)
ref = None
for rnd in range(2):
    out = []
    for sp in (1, 2, 3, 4, 6, 8):
        L.check(L.lib().a8_set_tuning(6, sp), "tune")
        dw = torch.zeros((G, d // G, k, d // G), device="cuda")
        K.posconv_wgrad(x, dz, dw, G, k, 63)
        if ref is None:
            ref = dw.clone()
        assert torch.allclose(dw, ref, rtol=1e-3, atol=1e-2), sp
        out.append("%d:%.1f" % (sp, timeit(lambda: K.posconv_wgrad(x, dz, dw, G, k, 63), iters=20)))
    print("round", rnd, " ".join(out))
