#!/usr/bin/env python3
"""Race screen of the stream GEMM (csrc/gemm_stream.hip): a new DMA-ring / barrier structure must be screened over many runs at
several sizes, under uneven load (cdna guide, 'A sync-structure edit makes a NEW template').  Random shapes, tile heights, ring
depths and epilogues; every output compared bit for bit with the 128 x 128 kernel's; a second stream keeps the memory system and
some CUs busy with copies and small GEMMs of its own while the stream kernel runs.
usage: stream_soak.py [iterations]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import _lib as L, kernels as K  # noqa: E402

it = int(sys.argv[1]) if len(sys.argv) > 1 else 150
dev, bf = "cuda", torch.bfloat16
rng = random.Random(1234)
g = torch.Generator(device="cpu").manual_seed(7)


def tune(k, v):
    L.check(L.lib().a8_set_tuning(k, v), "tune")


noise_a = torch.randn(64 << 20, device=dev, dtype=bf)
noise_b = torch.empty_like(noise_a)
na, nb = torch.randn(3000, 768, device=dev, dtype=bf), torch.randn(768, 768, device=dev, dtype=bf)
side = torch.cuda.Stream()
bad = 0
for i in range(it):
    K_ = rng.choice([256, 384, 512, 640, 768, 1024, 1536, 2304])
    N = rng.choice([256, 512, 768, 1024])
    nbatch = rng.choice([1, 1, 2, 3, 8])
    rpb = rng.randint(160, 40000 // nbatch + 200)
    M = nbatch * rpb
    bm = rng.choice([0, 160, 192, 224, 256])
    mode = rng.choice(["plain", "bias", "gelu_z", "mulz", "addz"])
    A = (torch.randn(M, K_, generator=g) * 0.5).to(bf).to(dev)
    B = (torch.randn(N, K_, generator=g) * 0.5).to(bf).to(dev)
    bias = torch.randn(N, device=dev) if mode in ("bias", "gelu_z") else None
    z = torch.randn(M, N, device=dev).to(bf) if mode in ("mulz", "addz") else None

    def run():
        kw = {}
        if bias is not None:
            kw["bias"] = bias
        if mode == "gelu_z":
            zo = torch.empty(M, N, device=dev, dtype=bf)
            return K.gemm(L.NT, A, B, M, N, K_, act=L.ACT_GELU, zout=zo, **kw), zo
        if mode == "mulz":
            return (K.gemm(L.NT, A, B, M, N, K_, act=L.ACT_MUL_GELU_GRAD, zin=z),)
        if mode == "addz":
            return (K.gemm(L.NT, A, B, M, N, K_, act=L.ACT_ADD, zin=z),)
        return (K.gemm(L.NT, A, B, M, N, K_, **kw),)

    tune(5, 0)
    tune(14, 1)                                        # every problem on the asymmetric-ring kernel ...
    tune(22, 1)                                        # ... with the LDS-staged epilogue: the reference
    ref = run()
    torch.cuda.synchronize()
    # round 5: the same kernel with the LDS-free epilogue (no barrier between the main loop and the stores) under the same uneven load
    tune(22, 0)
    with torch.cuda.stream(side):
        for _ in range(rng.randint(0, 2)):
            noise_b.copy_(noise_a)
            K.gemm(L.NT, na, nb, 3000, 768, 768)
    got_d = [run() for _ in range(2)]
    torch.cuda.synchronize()
    for gg in got_d:
        for r, o in zip(ref, gg):
            if not torch.equal(r, o):
                bad += 1
                print("MISMATCH (direct epilogue) it %d M %d N %d K %d %s" % (i, M, N, K_, mode), flush=True)
    tune(5, 1)
    tune(19, 1)                                        # plain (not row-remapped) problems through the stream kernel too
    tune(7, bm)
    with torch.cuda.stream(side):                      # uneven load beside the kernel under test
        for _ in range(rng.randint(0, 3)):
            noise_b.copy_(noise_a)
            K.gemm(L.NT, na, nb, 3000, 768, 768)
    got = [run() for _ in range(2)]                    # twice: back to back, the second with warm caches
    torch.cuda.synchronize()
    for gg in got:
        for r, o in zip(ref, gg):
            if not torch.equal(r, o):
                bad += 1
                d = (r.float() - o.float()).abs()
                print("MISMATCH it %d M %d (rpb %d) N %d K %d bm %d %s: %d elements, max %.4g, first row %d" % (
                    i, M, rpb, N, K_, bm, mode, int((d > 0).sum()), float(d.max()), int((d > 0).any(1).nonzero()[0])), flush=True)
    if i % 25 == 0:
        print("it %d ok so far (bad %d)" % (i, bad), flush=True)
tune(5, 20000)
tune(7, 0)
tune(19, 0)
tune(14, 1000)
print("stream soak: %d iterations, %d mismatching outputs" % (it, bad))
sys.exit(1 if bad else 0)
