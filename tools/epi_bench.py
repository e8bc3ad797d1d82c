#!/usr/bin/env python3
"""What the GEMM epilogue variants cost on the FFN1 and conv1 shapes (interleaved rounds, one process)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import _lib as L, kernels as K  # noqa: E402
from tools.micro_bench import timeit  # noqa: E402


def main():
    dev, bf = "cuda", torch.bfloat16
    for name, M, N, Kd, extra in (("ffn1", 6248, 3072, 768, {}), ("qkv", 6248, 2304, 768, {}), ("o", 6248, 768, 768, {}),
                                  ("conv1", 8 * 24999, 512, 1536, dict(lda=1024, a_rpb=24999, a_bs=49999 * 512))):
        A = torch.randn(8 * 49999 * 512 if extra else M * Kd, device=dev).to(bf)
        B = torch.randn(N, Kd, device=dev).to(bf)
        bias = torch.randn(N, device=dev)
        z = torch.randn(M, N, device=dev).to(bf)
        out, out2 = torch.empty(M, N, device=dev, dtype=bf), torch.empty(M, N, device=dev, dtype=bf)
        cs = torch.zeros(N, device=dev)
        variants = {
            "plain": dict(),
            "bias": dict(bias=bias),
            "bias+gelu": dict(bias=bias, act=L.ACT_GELU),
            "bias+gelu+zout": dict(bias=bias, act=L.ACT_GELU, zout=out2),
            "mul_gelu_grad": dict(act=L.ACT_MUL_GELU_GRAD, zin=z),
            "mul_gelu_grad+colsum": dict(act=L.ACT_MUL_GELU_GRAD, zin=z, colsum=cs),
            "add": dict(act=L.ACT_ADD, zin=z),
        }
        for rnd in range(2):
            res = []
            for vn, kw in variants.items():
                t = timeit(lambda: K.gemm(L.NT, A, B, M, N, Kd, out=out, **extra, **kw), iters=30)
                res.append("%s %.1f" % (vn, t))
            print(name, "round", rnd, " | ".join(res))


if __name__ == "__main__":
    main()
