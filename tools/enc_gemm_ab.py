#!/usr/bin/env python3
"""Interleaved A/B of the encoder's bf16 NT products (the shapes and epilogues a wav2vec2-base layer launches, whole batch M = 6248 and
half batch M = 3124) under a8_set_tuning variants, alone on the GPU (HIP events, 25 launches per variant per round, 5 rounds).
usage: enc_gemm_ab.py "name:key=val,key=val;name2:..."   e.g.  "direct:;lds:22=1" """
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import _lib as L, kernels as K  # noqa: E402

DEFAULTS = {5: 20000, 7: 0, 8: 0, 14: 1000, 15: 1, 19: 0, 22: 0, 23: 0, 25: 0}      # (key 24 selected the round-5 wide-tile kernel, which lives only in tools/ubench/round5_gemm_wide: no library code reads it)
cfgs = []
for part in (sys.argv[1] if len(sys.argv) > 1 else "base:").split(";"):
    name, _, kv = part.partition(":")
    cfgs.append((name, {int(k): int(v) for k, v in (it.split("=") for it in filter(None, kv.split(",")))}))
dev, bf = "cuda", torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)


def rnd(*shape):
    return torch.randn(*shape, device=dev, generator=g).to(bf)


def problems(M):
    d, dff = 768, 3072
    x, h, dq = rnd(M, d), rnd(M, dff), rnd(M, 3 * d)
    wqkv, wo, w1, w2 = rnd(3 * d, d), rnd(d, d), rnd(dff, d), rnd(d, dff)
    bq, bo, b1, b2 = (torch.randn(n, device=dev) for n in (3 * d, d, dff, d))
    z1, res, wqkv_t = rnd(M, dff), rnd(M, d), rnd(d, 3 * d)
    zo = torch.empty(M, dff, device=dev, dtype=bf)
    o3, o1, of = torch.empty(M, 3 * d, device=dev, dtype=bf), torch.empty(M, d, device=dev, dtype=bf), torch.empty(M, dff, device=dev, dtype=bf)
    return [
        ("qkv  fwd  N 2304 K  768 bias", 2.0 * M * 2304 * 768, lambda: K.gemm(L.NT, x, wqkv, M, 3 * d, d, bias=bq, out=o3)),
        ("o    fwd  N  768 K  768 bias", 2.0 * M * 768 * 768, lambda: K.gemm(L.NT, x, wo, M, d, d, bias=bo, out=o1)),
        ("ffn1 fwd  N 3072 K  768 bias+gelu+z", 2.0 * M * 3072 * 768, lambda: K.gemm(L.NT, x, w1, M, dff, d, bias=b1, act=L.ACT_GELU, zout=zo, out=of)),
        ("ffn2 fwd  N  768 K 3072 bias", 2.0 * M * 768 * 3072, lambda: K.gemm(L.NT, h, w2, M, d, dff, bias=b2, out=o1)),
        ("dffn2     N 3072 K  768 x gelu'(z)", 2.0 * M * 3072 * 768, lambda: K.gemm(L.NT, x, w1, M, dff, d, act=L.ACT_MUL_GELU_GRAD, zin=z1, out=of)),
        ("dffn1     N  768 K 3072 + res", 2.0 * M * 768 * 3072, lambda: K.gemm(L.NT, h, w2, M, d, dff, act=L.ACT_ADD, zin=res, out=o1)),
        ("do        N  768 K  768", 2.0 * M * 768 * 768, lambda: K.gemm(L.NT, x, wo, M, d, d, out=o1)),
        ("dqkv      N  768 K 2304 + res", 2.0 * M * 768 * 2304, lambda: K.gemm(L.NT, dq, wqkv_t, M, d, 3 * d, act=L.ACT_ADD, zin=res, out=o1)),
    ]


def apply(dct):
    for k, v in DEFAULTS.items():
        L.check(L.lib().a8_set_tuning(k, dct.get(k, v)), "tune")


reps, rounds = int(os.environ.get("AB_REPS", "25")), int(os.environ.get("AB_ROUNDS", "5"))
for M in (6248, 3124):
    probs = problems(M)
    print(f"== M = {M}")
    tot = {n: 0.0 for n, _ in cfgs}
    for pname, flops, fn in probs:
        res = {n: [] for n, _ in cfgs}
        for _ in range(rounds):
            for name, dct in cfgs:
                apply(dct)
                for _ in range(3):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res[name].append(e0.elapsed_time(e1) * 1e3 / reps)
        line = f"  {pname:38s}"
        for name, _ in cfgs:
            med = float(np.median(res[name]))
            tot[name] += med
            line += f"  {name} {med:6.1f} us ({flops / med / 1e6:5.0f} TF/s)"
        print(line, flush=True)
    print("  sum over the eight products:        " + "  ".join(f"{n} {t:6.1f} us" for n, t in tot.items()))
apply({})
