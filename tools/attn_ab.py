#!/usr/bin/env python3
"""Attention launch-geometry / register-cap A/B at the cfg2 shape (interleaved rounds, one process).
a8_set_tuning keys 2 (fwd), 3 (dK/dV), 4 (dQ): value = waves*10 + groups (141: the default geometry with run-time flags)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import _lib as L, kernels as K  # noqa: E402
from tools.micro_bench import timeit  # noqa: E402

B, T, H = 8, 781, 12
dm = H * 64
bf = torch.bfloat16
qkv = (torch.randn(B * T, 3 * dm, device="cuda") * 0.5).to(bf)
dout = torch.randn(B * T, dm, device="cuda").to(bf)
p = 0.1
o, lse = K.attn_fwd(qkv[:, :dm], qkv[:, dm:2 * dm], qkv[:, 2 * dm:], B, T, H, 3 * dm, 0.125, None, p=p, seed=3)
dq = torch.empty(B * T, 3 * dm, device="cuda", dtype=bf)


def tune(k, v):
    L.check(L.lib().a8_set_tuning(k, v), "tune")


def fwd():
    return K.attn_fwd(qkv[:, :dm], qkv[:, dm:2 * dm], qkv[:, 2 * dm:], B, T, H, 3 * dm, 0.125, None, p=p, seed=3)


def bwd():
    return K.attn_bwd(qkv[:, :dm], qkv[:, dm:2 * dm], qkv[:, 2 * dm:], o, dout, lse, B, T, H, 3 * dm, 0.125, None, p=p, seed=3, dqkv=dq)


tune(2, 41); tune(3, 41); tune(4, 41)
ref_o = fwd()[0].clone()
ref_lse = fwd()[1].clone()
ref_d = bwd()[3].clone()
for rnd in range(3):
    out = []
    for v in (41, 0, 4, 5):                     # 41: the 16x16 kernel (4 waves x 16 queries); 0 / 4 / 5: the 32x32 kernel (auto / 4 / 5 waves x 32 queries)
        tune(2, v)
        got, lse_ = fwd()[:2]
        err = (got.float() - ref_o.float()).abs().max().item()
        assert err < 2e-2 and (lse_ - ref_lse).abs().max().item() < 1e-3, (v, err)
        out.append("fwd %d: %.1f (max |d| %.1e)" % (v, timeit(fwd, iters=30), err))
    tune(2, 0)
    for kq in (41, 32):           # 41: the 16 x 16 kernels (dropout / mask flags compiled in); 32: the 32 x 32-MFMA kernels of round 6
        for kk in (41, 32):
            tune(4, kq); tune(3, kk)
            err = (bwd()[3].float() - ref_d.float()).abs().max().item()
            assert err < 3e-2, (kq, kk, err)
            out.append("bwd dq=%d kv=%d: %.1f (max |d| %.1e)" % (kq, kk, timeit(bwd, iters=30), err))
    tune(3, 41); tune(4, 41)
    print("round %d | " % rnd + " | ".join(out), flush=True)
