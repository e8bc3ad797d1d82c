#!/usr/bin/env python3
"""ln_bwd geometry sweep (a8_set_tuning key 1 = waves_per_block*100 + rows_per_wave), interleaved rounds in one process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import _lib as L, kernels as K  # noqa: E402
from tools.micro_bench import timeit  # noqa: E402


def main():
    d, bf = "cuda", torch.bfloat16
    rows, cols = 6248, 768
    x, res, dy, dy2 = (torch.randn(rows, cols, device=d).to(bf) for _ in range(4))
    g, b = torch.ones(cols, device=d), torch.zeros(cols, device=d)
    dg, db = torch.zeros(cols, device=d), torch.zeros(cols, device=d)
    y, mean, rstd = K.layernorm_fwd(x, res, g, b, 1e-6, 0.1, 1, 1)
    variants = [403, 401, 402, 801, 802, 1601, 1602, 1603]
    ref = None
    for v in variants:
        L.check(L.lib().a8_set_tuning(1, v), "tune")
        dg.zero_(); db.zero_()
        dx, dres = K.layernorm_bwd(dy, dy2, x, res, g, mean, rstd, dg, db, 0.1, 1, 1, want_dx=True, want_dres=True)
        cur = (dx.float().clone(), dg.clone(), db.clone())
        if ref is None:
            ref = cur
        else:
            assert torch.equal(cur[0], ref[0])
            assert torch.allclose(cur[1], ref[1], rtol=1e-4, atol=1e-2) and torch.allclose(cur[2], ref[2], rtol=1e-4, atol=1e-2)
    for rnd in range(3):
        out = []
        for v in variants:
            L.check(L.lib().a8_set_tuning(1, v), "tune")
            t = timeit(lambda: K.layernorm_bwd(dy, dy2, x, res, g, mean, rstd, dg, db, 0.1, 1, 1, want_dx=True, want_dres=True), iters=50)
            out.append("%d:%.1f" % (v, t))
        print("round", rnd, " ".join(out))
    import ctypes as C
    for v in variants:
        L.check(L.lib().a8_set_tuning(1, v), "tune")
        def f():
            L.check(L.lib().a8_layernorm_bwd(L.dt(x), L.ptr(dy), L.ptr(dy2), L.ptr(x), L.ptr(res), L.ptr(g), L.ptr(mean), L.ptr(rstd), L.ptr(dx), L.ptr(dres),
                                             None, None, None, rows, cols, 0.1, 1, 1, L.stream()), "ln")
        print("no-colsum %d: %.1f us" % (v, timeit(f, iters=50)))


if __name__ == "__main__":
    main()
