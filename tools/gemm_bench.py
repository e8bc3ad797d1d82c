#!/usr/bin/env python3
"""Time a8_gemm on the shapes of the wav2vec2-base step (bf16) for each stage-geometry variant.
Usage (GPU box): python tools/gemm_bench.py [variants...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import _lib as L, kernels as K  # noqa: E402

M = 6248
SHAPES = [  # name, layout, M, N, K, kwargs
    ("qkv   NT", L.NT, M, 2304, 768, {}),
    ("o     NT", L.NT, M, 768, 768, {}),
    ("ffn1  NT", L.NT, M, 3072, 768, {}),
    ("ffn2  NT", L.NT, M, 768, 3072, {}),
    ("dz1   NN", L.NN, M, 3072, 768, {}),
    ("dx1   NN", L.NN, M, 768, 3072, {}),
    ("do    NN", L.NN, M, 768, 768, {}),
    ("dW1   TN", L.TN, 3072, 768, M, {"splitk": 3}),
    ("dW2   TN", L.TN, 768, 3072, M, {"splitk": 3}),
    ("dWo   TN", L.TN, 768, 768, M, {"splitk": 14}),
    ("dWo/7 TN", L.TN, 768, 768, M, {"splitk": 7}),
    ("dWqkv TN", L.TN, 2304, 768, M, {"splitk": 4}),
    ("conv1 NT", L.NT, 8 * 24999, 512, 1536, {"conv": (49999, 24999, 3)}),
    ("conv1 NN", L.NN, 8 * 24999, 1536, 512, {}),
    ("conv1 TN", L.TN, 512, 1536, 8 * 24999, {"splitk": 10, "convb": (49999, 24999, 3)}),
    ("conv5 NT", L.NT, 8 * 1562, 512, 1024, {"conv": (3124, 1562, 2)}),
]


def run(variant, iters=20):
    # (the stage-geometry variants of round 2 no longer ship: `variant` only labels the run)
    dev = "cuda"
    out = []
    for name, layout, m, n, k, kw in SHAPES:
        kw = dict(kw)
        gk = {}
        if "conv" in kw:
            Lin, Lout, kk = kw.pop("conv")
            A = torch.randn(8 * Lin * 512, device=dev).to(torch.bfloat16)
            gk = dict(lda=1024, a_rpb=Lout, a_bs=Lin * 512)
            B = torch.randn(n, k, device=dev).to(torch.bfloat16)
        elif "convb" in kw:
            Lin, Lout, kk = kw.pop("convb")
            A = torch.randn(k, m, device=dev).to(torch.bfloat16)
            B = torch.randn(8 * Lin * 512, device=dev).to(torch.bfloat16)
            gk = dict(ldb=1024, b_rpb=Lout, b_bs=Lin * 512)
        elif layout == L.NT:
            A, B = torch.randn(m, k, device=dev).to(torch.bfloat16), torch.randn(n, k, device=dev).to(torch.bfloat16)
        elif layout == L.NN:
            A, B = torch.randn(m, k, device=dev).to(torch.bfloat16), torch.randn(k, n, device=dev).to(torch.bfloat16)
        else:
            A, B = torch.randn(k, m, device=dev).to(torch.bfloat16), torch.randn(k, n, device=dev).to(torch.bfloat16)
        odt = torch.float32 if layout == L.TN else torch.bfloat16
        C = torch.zeros(m, n, device=dev, dtype=odt)
        sk = kw.get("splitk", 1)
        for _ in range(3):
            K.gemm(layout, A, B, m, n, k, out=C, splitk=sk, accumulate=sk > 1, **gk)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            K.gemm(layout, A, B, m, n, k, out=C, splitk=sk, accumulate=sk > 1, **gk)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        out.append((name, us, 2.0 * m * n * k / us / 1e6))
    return out


if __name__ == "__main__":
    variants = [int(v) for v in sys.argv[1:]] or [0, 1, 2, 3]
    res = {v: run(v) for v in variants}
    print("%-10s" % "shape" + "".join("   v%d: us    TF/s" % v for v in variants))
    for i, (name, _, _) in enumerate(res[variants[0]]):
        print("%-10s" % name + "".join("  %8.1f %7.1f" % (res[v][i][1], res[v][i][2]) for v in variants))
    for v in variants:
        print("variant", v, "sum us", round(sum(r[1] for r in res[v]), 1))
