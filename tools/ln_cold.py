#!/usr/bin/env python3
"""ln_bwd as the encoder layer calls it (dgamma, dbeta, dbias; dx + dres), on rotating buffer sets so that every call
reads cold operands (the in-step situation), against the same call on one hot set."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import _lib as L, kernels as K  # noqa: E402
from tools.micro_bench import timeit  # noqa: E402


def main():
    d, bf = "cuda", torch.bfloat16
    rows, cols = 6248, 768
    nset = int(os.environ.get("NSET", "16"))
    sets = []
    g, b = torch.ones(cols, device=d), torch.zeros(cols, device=d)
    for _ in range(nset):
        x, res, dy, dy2 = (torch.randn(rows, cols, device=d).to(bf) for _ in range(4))
        y, mean, rstd = K.layernorm_fwd(x, res, g, b, 1e-6, 0.1, 1, 1)
        dx, dres = torch.empty_like(x), torch.empty_like(x)
        sets.append((x, res, dy, dy2, mean, rstd, dx, dres, torch.zeros(3, cols, device=d)))
    variants = [int(v) for v in os.environ.get("VARIANTS", "403,802,1601,1602").split(",")]
    lib = L.lib()

    def call(s, colsum, dbias, dy2):
        x, res, dy, dy2_, mean, rstd, dx, dres, acc = s
        L.check(lib.a8_layernorm_bwd(L.dt(x), L.ptr(dy), L.ptr(dy2_ if dy2 else None), L.ptr(x), L.ptr(res), L.ptr(g), L.ptr(mean), L.ptr(rstd),
                                     L.ptr(dx), L.ptr(dres), L.ptr(acc[0]) if colsum else None, L.ptr(acc[1]) if colsum else None,
                                     L.ptr(acc[2]) if dbias else None, rows, cols, 0.1, 1, 1, L.stream()), "ln")

    for v in variants:
        L.check(lib.a8_set_tuning(1, v), "tune")
        for colsum, dbias, dy2 in ((0, 0, 0), (1, 0, 0), (1, 1, 0), (1, 1, 1)):
            k = [0]

            def cold():
                call(sets[k[0] % nset], colsum, dbias, dy2)
                k[0] += 1
            t_cold = timeit(cold, iters=4 * nset)
            t_hot = timeit(lambda: call(sets[0], colsum, dbias, dy2), iters=50)
            print("geom %4d colsum %d dbias %d dy2 %d: cold %.1f us  hot %.1f us" % (v, colsum, dbias, dy2, t_cold, t_hot), flush=True)


if __name__ == "__main__":
    main()
