#!/usr/bin/env python3
"""Interleaved A/B of the weight-gradient (TN) products alone on the GPU under a8_set_tuning variants: the encoder layer's grouped launch
(dW1, dW2, dWo, dWqkv at M = 6248) and the conv layers' split-K products with their batch remaps.
usage: wgrad_ab.py "name:key=val,...;name2:..." """
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import _lib as L, kernels as K, ops  # noqa: E402

DEFAULTS = {14: 1000, 15: 1, 25: 0}
cfgs = []
for part in (sys.argv[1] if len(sys.argv) > 1 else "base:").split(";"):
    name, _, kv = part.partition(":")
    cfgs.append((name, {int(k): int(v) for k, v in (it.split("=") for it in filter(None, kv.split(",")))}))
dev, bf = "cuda", torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g).to(bf)


def apply(d):
    for k, v in DEFAULTS.items():
        L.check(L.lib().a8_set_tuning(k, d.get(k, v)), "tune")


M, d, dff = 6248, 768, 3072
shapes = [(d, dff), (dff, d), (d, d), (3 * d, d)]
opsd = [(rnd(M, a), rnd(M, b)) for a, b in shapes]
tg = [torch.zeros(a, b, device=dev) for a, b in shapes]
probs = [("encoder layer, grouped (4 products, 88.9 GF)", 2.0 * M * sum(a * b for a, b in shapes),
          lambda: K.gemm_grouped(L.TN, [(dy, x, a, b, M, dict(out=t, accumulate=True)) for (a, b), (dy, x), t in zip(shapes, opsd, tg)]))]
Bn = 8
for li, (Lin, k, s_) in enumerate([(49999, 3, 2), (24999, 3, 2), (12499, 3, 2), (6249, 3, 2), (3124, 2, 2), (1562, 2, 2)], 1):
    Lout = (Lin - k) // s_ + 1
    co = cin = 512
    pad = 1
    xin = rnd(Bn, Lin, cin)
    dz = rnd(Bn, Lout + 2 * pad, co)
    dw = torch.zeros(co, k * cin, device=dev)
    sk = ops._splitk((co // 128) * (k * cin // 128), Bn * Lout)

    def run(dz=dz, xin=xin, dw=dw, Lout=Lout, Lin=Lin, k=k, s_=s_, sk=sk):
        K.gemm(L.TN, dz, xin, co, k * cin, Bn * Lout, out=dw, splitk=sk, ldb=s_ * cin, b_rpb=Lout, b_bs=Lin * cin, lda=co,
               a_off=pad * co, a_rpb=Lout, a_bs=(Lout + 2 * pad) * co)
    probs.append(("conv layer %d wgrad (rows %d, split-K asked %d)" % (li, Bn * Lout, sk), 2.0 * Bn * Lout * co * k * cin, run))

reps, rounds = 20, 5
tot = {n: 0.0 for n, _ in cfgs}
for pname, flops, fn in probs:
    res = {n: [] for n, _ in cfgs}
    for _ in range(rounds):
        for name, dct in cfgs:
            apply(dct)
            for _ in range(2):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) * 1e3 / reps)
    line = f"  {pname:52s}"
    for name, _ in cfgs:
        med = float(np.median(res[name]))
        tot[name] += med
        line += f"  {name} {med:7.1f} us ({flops / med / 1e6:5.0f} TF/s)"
    print(line, flush=True)
print("  sum: " + "  ".join(f"{n} {t:7.1f} us" for n, t in tot.items()))
apply({})
