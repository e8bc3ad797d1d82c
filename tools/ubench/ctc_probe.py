#!/usr/bin/env python3
"""a8_ctc_fwd_bwd alone at the cfg5 shape (16 utterances, 624 frames, 32 labels, 50..300 targets) and two more target lengths: HIP-event time per call,
loss and gradient checksums (same inputs every run), and the maximum deviation from torch's own ctc_loss (f64 on the CPU) for the first shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from audio8_amd import kernels as K  # noqa: E402


def case(B, T, V, smin, smax, check):
    g = torch.Generator().manual_seed(0)
    lp = torch.log_softmax(2.0 * torch.randn(T, B, V, generator=g), -1)
    in_len = ((torch.rand(B, generator=g) * 0.4 + 0.6) * T).int()
    in_len[0] = T
    tl = torch.randint(smin, smax + 1, (B,), generator=g, dtype=torch.int32)
    tl[1] = smax
    flat = torch.cat([torch.randint(1, V, (int(n),), generator=g, dtype=torch.int32) for n in tl])
    off = torch.zeros(B + 1, dtype=torch.int32)
    off[1:] = torch.cumsum(tl, 0)
    d = "cuda"
    args = (lp.to(d), flat.to(d), off.to(d), in_len.to(d), int(tl.max()))
    nll, grad = K.ctc_fwd_bwd(*args)
    torch.cuda.synchronize()
    n = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        K.ctc_fwd_bwd(*args)
    e1.record()
    torch.cuda.synchronize()
    msg = "B %d T %d V %d targets %d..%d: %7.1f us per call; sum nll %.6f  sum|grad| %.6f" % (
        B, T, V, smin, smax, e0.elapsed_time(e1) * 1e3 / n, nll.double().sum().item(), grad.double().abs().sum().item())
    if check:
        lp64 = lp.double().requires_grad_(True)
        ref = torch.nn.functional.ctc_loss(lp64, flat.long(), in_len.long(), tl.long(), blank=0, reduction="none", zero_infinity=True)
        ref.sum().backward()
        mask = (torch.arange(T).unsqueeze(1) < in_len.unsqueeze(0)).unsqueeze(-1)
        msg += "; vs torch f64: nll %.2e rel, grad %.2e abs" % (((nll.cpu().double() - ref.detach()).abs() / ref.detach().abs().clamp_min(1)).max().item(),
                                                             ((grad.cpu().double() - lp64.grad) * mask).abs().max().item())
    print(msg)


case(16, 624, 32, 50, 300, True)
case(16, 624, 32, 20, 60, False)
case(8, 999, 32, 100, 480, False)
