#!/usr/bin/env python3
"""Debug aid for the 32x32 attention forward kernel: small shapes, no dropout, error structure against an fp64 reference."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import _lib as L, kernels as K  # noqa: E402

torch.manual_seed(0)
bf = torch.bfloat16
for (B, H, T, nw) in [(1, 1, 64, 4), (1, 1, 32, 4), (1, 1, 128, 4), (1, 2, 200, 5), (2, 3, 781, 5)]:
    d = H * 64
    qkv = (torch.randn(B * T, 3 * d) * 0.5).to(bf)
    f = qkv.double().view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    sc = torch.matmul(f[0], f[1].transpose(-2, -1)) * 0.125
    ref = torch.matmul(sc.softmax(-1), f[2]).permute(0, 2, 1, 3).reshape(B * T, d)
    g = qkv.cuda()
    for key2 in (41, nw):
        L.check(L.lib().a8_set_tuning(2, key2), "tune")
        o, lse = K.attn_fwd(g[:, :d], g[:, d:2 * d], g[:, 2 * d:], B, T, H, 3 * d, 0.125, None)
        torch.cuda.synchronize()
        e = (o.double().cpu() - ref).abs()
        lse_ref = (torch.logsumexp(sc, -1) * 1.4426950408889634)
        le = (lse.double().cpu().view(B, H, T) - lse_ref).abs().max().item()
        print("B%d H%d T%d key2=%d: max |o - ref| %.3e  lse err %.3e  nan %d" % (B, H, T, key2, e.max().item(), le, int(torch.isnan(o).sum())), flush=True)
        if key2 == nw and e.max().item() > 5e-2:
            eq = e.view(B, T, H, 64)
            print("  worst per query (first 40):", [round(x, 2) for x in eq[0, :40, 0].max(-1).values.tolist()])
            print("  worst per d (head 0):", [round(x, 2) for x in eq[0, :, 0].max(0).values.tolist()])
            # which keys contribute wrongly: V = one-hot probes
            v1 = torch.zeros(B, T, H, 64)
            n = min(64, T)
            v1[:, :n] = torch.eye(64)[:n, None, :]
            qz = torch.zeros(B * T, d)
            o2, _ = K.attn_fwd(qz.to(bf).cuda(), qz.to(bf).cuda(), v1.reshape(B * T, d).to(bf).cuda(), B, T, H, d, 0.125, None)
            print("  uniform-P probe, query 0, head 0 (want 1/T = %.4f on the first %d dims):" % (1.0 / T, n), [round(x, 4) for x in o2.float().cpu().view(B, T, H, 64)[0, 0, 0].tolist()])
            print("  uniform-P probe, query 1:", [round(x, 4) for x in o2.float().cpu().view(B, T, H, 64)[0, 1, 0].tolist()])
L.check(L.lib().a8_set_tuning(2, 0), "tune")
