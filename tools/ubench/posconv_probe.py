#!/usr/bin/env python3
"""The positional-conv kernels alone at the cfg2 (8 x 781 x 768, 48 channels per group) and cfg4 (4 x 999 x 1024, 64 per group) shapes, bf16:
HIP-event time per launch and checksums of the results (the same operands every run: two builds of a kernel print the same sums up to the
f32 summation order).  Under `rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python3 tools/ubench/posconv_probe.py`
the LDS counters per launch (profiles/r06_posconv_lds.txt)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from audio8_amd import _lib as L, kernels as K  # noqa: E402

L.check(L.lib().a8_set_tuning(6, 3), "tune")           # the weight gradient's grid split as the step runs it (a8_set_tuning key 6)


def timed(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for name, (B, T, d, groups) in (("cfg2", (8, 781, 768, 16)), ("cfg4", (4, 999, 1024, 16))):
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(B, T, d, device="cuda", generator=g).to(torch.bfloat16)
    dz = (0.1 * torch.randn(B, T, d, device="cuda", generator=g)).to(torch.bfloat16)
    dg = d // groups
    v = 0.05 * torch.randn(d, dg, 128, device="cuda", generator=g)
    gain = torch.rand(1, 1, 128, device="cuda", generator=g) + 0.5
    bias = 0.1 * torch.randn(d, device="cuda", generator=g)
    wf, wt, _ = K.weightnorm_pack(v, gain, groups, torch.bfloat16)
    y, z = K.posconv_fwd(x, wf, bias, groups, 128, 64, 1, want_z=True)
    us = timed(lambda: K.posconv_fwd(x, wf, bias, groups, 128, 64, 1, want_z=True))
    print("%s posconv_fwd   bf16 dg=%d: %6.1f us per launch; sum y %.6e  sum|z| %.6e" % (name, dg, us, y.double().sum().item(), z.double().abs().sum().item()))
    dw = torch.zeros(groups * dg, 128, dg, device="cuda")
    K.posconv_wgrad(x, dz, dw, groups, 128, 64)
    torch.cuda.synchronize()
    ref = dw.clone()
    us = timed(lambda: K.posconv_wgrad(x, dz, dw, groups, 128, 64))
    print("%s posconv_wgrad bf16 dg=%d: %6.1f us per launch; sum %.6e  sum|.| %.6e" % (name, dg, us, ref.double().sum().item(), ref.double().abs().sum().item()))
