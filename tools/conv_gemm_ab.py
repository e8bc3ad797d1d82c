#!/usr/bin/env python3
"""The conv-layer GEMMs (N = 512) under the default 128 x 128 kernel vs the alternatives, interleaved rounds in one process:
KEY=11 (full-width persistent 128 x 512), KEY=5 (pipelined 256 x 256).  Forward shape = overlapping rows + GELU + zout."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import _lib as L, kernels as K  # noqa: E402
from tools.micro_bench import timeit  # noqa: E402

bf = torch.bfloat16
cfgs = [("default", {}), ("a3", {14: 1}), ("w512", {11: 1}), ("p256", {5: 1})]
if os.environ.get("AB_ONLY"):
    cfgs = [c for c in cfgs if c[0] in ("default",) + tuple(os.environ["AB_ONLY"].split(","))]
B, C = 8, 512
layers = [(49999, 24999, 3), (24999, 12499, 3), (12499, 6249, 3), (6249, 3124, 3), (3124, 1562, 2), (1562, 781, 2)]


def tune(d):
    for k in (5, 11, 14):
        L.check(L.lib().a8_set_tuning(k, d.get(k, 0)), "tune")


tot = {n: 0.0 for n, _ in cfgs}
for Lin, Lout, k in layers:
    x = torch.randn(B * Lin * C, device="cuda").to(bf)
    W = (torch.randn(C, k * C, device="cuda") * (k * C) ** -0.5).to(bf)
    M = B * Lout
    y, z = torch.empty(M, C, device="cuda", dtype=bf), torch.empty(M, C, device="cuda", dtype=bf)
    dz = torch.randn(M, C, device="cuda").to(bf)
    Wt = (torch.randn(k * C, C, device="cuda") * C ** -0.5).to(bf)       # [k*cin, co]: dcol = dz . W (as an NT product on W^T)
    dcol = torch.empty(M, k * C, device="cuda", dtype=bf)
    res, ref = [], None
    for rnd in range(3):
        for name, d in cfgs:
            tune(d)
            tf = timeit(lambda: K.gemm(L.NT, x, W, M, C, k * C, lda=2 * C, a_rpb=Lout, a_bs=Lin * C, act=L.ACT_GELU, zout=z, out=y), iters=10)
            if ref is None:
                ref = (y.clone(), z.clone())
            else:
                assert torch.equal(ref[0], y) and torch.equal(ref[1], z), name
            tb = timeit(lambda: K.gemm(L.NT, dz, Wt, M, k * C, C, out=dcol), iters=10)
            if rnd == 2:
                fl = 2.0 * M * C * k * C
                res.append("%s fwd %.1f us (%.0f TF) dgrad-like %.1f us (%.0f TF)" % (name, tf, fl / tf / 1e6, tb, fl / tb / 1e6))
                tot[name] += tf
    print("conv Lout=%d k=%d: " % (Lout, k) + " | ".join(res), flush=True)
print("forward totals (us):", {k: round(v, 1) for k, v in tot.items()})
tune({})
