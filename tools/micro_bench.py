#!/usr/bin/env python3
"""Isolated timings of the row kernels at the wav2vec2-base shapes (GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import kernels as K  # noqa: E402


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    d = "cuda"
    rows, cols = 6248, 768
    bf = torch.bfloat16
    x, res, dy, dy2 = (torch.randn(rows, cols, device=d).to(bf) for _ in range(4))
    g, b = torch.ones(cols, device=d), torch.zeros(cols, device=d)
    dg, db = torch.zeros(cols, device=d), torch.zeros(cols, device=d)
    for p in (0.0, 0.1):
        y, mean, rstd = K.layernorm_fwd(x, res, g, b, 1e-6, p, 1, 1)
        print(f"ln_fwd p={p}: {timeit(lambda: K.layernorm_fwd(x, res, g, b, 1e-6, p, 1, 1)):.1f} us")
        print(f"ln_bwd p={p} (dy2, dx+dres): {timeit(lambda: K.layernorm_bwd(dy, dy2, x, res, g, mean, rstd, dg, db, p, 1, 1, want_dx=True, want_dres=p > 0)):.1f} us")
        print(f"ln_bwd p={p} (no dy2, dx only): {timeit(lambda: K.layernorm_bwd(dy, None, x, res, g, mean, rstd, dg, db, p, 1, 1, want_dx=True, want_dres=False)):.1f} us")
    z = torch.randn(rows, 3072, device=d).to(bf)
    dz = torch.randn(rows, 3072, device=d).to(bf)
    dbias = torch.zeros(3072, device=d)
    from audio8_amd import ops
    print(f"colsum 3072: {timeit(lambda: ops.colsum(dz, dbias)):.1f} us;  colsum 768: {timeit(lambda: ops.colsum(dy, dg)):.1f} us")
    print(f"dropout elementwise: {timeit(lambda: K.dropout(x, 0.1, 1, 2)):.1f} us; add: {timeit(lambda: K.add(x, res)):.1f} us")


def attention():
    from audio8_amd import _lib as L
    d = "cuda"
    B, T, H = 8, 781, 12
    dm = H * 64
    bf = torch.bfloat16
    qkv = (torch.randn(B * T, 3 * dm, device=d) * 0.5).to(bf)
    dout = torch.randn(B * T, dm, device=d).to(bf)
    gf = 4.0 * B * H * T * T * 64 / 1e9
    p = 0.1
    o, lse = K.attn_fwd(qkv[:, :dm], qkv[:, dm:2 * dm], qkv[:, 2 * dm:], B, T, H, 3 * dm, 0.125, None, p=p, seed=3)
    dq = torch.empty(B * T, 3 * dm, device=d, dtype=bf)
    for rnd in range(2):
        for geo in (41, 22, 42):
            L.check(L.lib().a8_set_tuning(2, geo), "tune")
            tf = timeit(lambda: K.attn_fwd(qkv[:, :dm], qkv[:, dm:2 * dm], qkv[:, 2 * dm:], B, T, H, 3 * dm, 0.125, None, p=p, seed=3))
            res = [f"fwd {tf:.1f} us ({gf / tf * 1e3:.0f} TF/s)"]
            for key, name in ((3, "dkv"), (4, "dq")):
                other = 4 if key == 3 else 3
                L.check(L.lib().a8_set_tuning(key, geo), "tune")
                L.check(L.lib().a8_set_tuning(other, 41), "tune")
                tb = timeit(lambda: K.attn_bwd(qkv[:, :dm], qkv[:, dm:2 * dm], qkv[:, 2 * dm:], o, dout, lse, B, T, H, 3 * dm, 0.125, None, p=p, seed=3, dqkv=dq))
                res.append(f"bwd with {name}={geo}, other=41: {tb:.1f} us")
            print(f"round {rnd} geometry {geo}: " + "; ".join(res))
    for key in (2, 3, 4):
        L.check(L.lib().a8_set_tuning(key, 41), "tune")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "attn1":        # a few launches for counter collection
        B, T, H = 8, 781, 12
        dm = H * 64
        qkv = (torch.randn(B * T, 3 * dm, device="cuda") * 0.5).to(torch.bfloat16)
        dout = torch.randn(B * T, dm, device="cuda").to(torch.bfloat16)
        for _ in range(3):
            o, lse = K.attn_fwd(qkv[:, :dm], qkv[:, dm:2 * dm], qkv[:, 2 * dm:], B, T, H, 3 * dm, 0.125, None, p=0.1, seed=3)
            K.attn_bwd(qkv[:, :dm], qkv[:, dm:2 * dm], qkv[:, 2 * dm:], o, dout, lse, B, T, H, 3 * dm, 0.125, None, p=0.1, seed=3)
        torch.cuda.synchronize()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "attn":
        attention()
        sys.exit(0)
    main()
