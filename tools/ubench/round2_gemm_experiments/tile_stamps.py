#!/usr/bin/env python3
"""Where a GEMM tile's time goes: per-block timestamps (a8_debug_set_stamps: entry, first DMAs issued, first k-tile landed, main loop
done, epilogue done; 100 MHz counter) for representative products of the step, launched alone."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import _lib as L, kernels as K  # noqa: E402

bf, dev = torch.bfloat16, "cuda"
M = 6248
stamps = torch.zeros(16384 * 8, device=dev, dtype=torch.int64)
shapes = [("o      6248x768x768", M, 768, 768, {}), ("ffn2   6248x768x3072", M, 768, 3072, {}), ("qkv    6248x2304x768", M, 2304, 768, {}),
          ("ffn1+gelu 6248x3072x768", M, 3072, 768, {"gelu": True}), ("conv1-like 200000x512x1536", 200000, 512, 1536, {})]
variants = [("", {})] if len(sys.argv) < 2 else [("", {})] + [(" [key %s=%s]" % tuple(a.split("=")), {int(a.split("=")[0]): int(a.split("=")[1])}) for a in sys.argv[1:]]
for name, m, n, k, opt in [(nm + vn, m, n, k, dict(o, tune=vt)) for (nm, m, n, k, o) in shapes for (vn, vt) in variants]:
    A, B = torch.randn(m, k, device=dev).to(bf), torch.randn(n, k, device=dev).to(bf)
    out, zo = torch.empty(m, n, device=dev, dtype=bf), torch.empty(m, n, device=dev, dtype=bf)
    bias = torch.randn(n, device=dev)
    kw = dict(out=out, bias=bias)
    if opt.get("gelu"):
        kw.update(act=L.ACT_GELU, zout=zo)
    for kk, vv in opt.get("tune", {}).items():
        L.check(L.lib().a8_set_tuning(kk, vv), "tune")
    for _ in range(3):
        K.gemm(L.NT, A, B, m, n, k, **kw)
    torch.cuda.synchronize()
    stamps.zero_()
    L.check(L.lib().a8_debug_set_stamps(stamps.data_ptr()), "stamps")
    K.gemm(L.NT, A, B, m, n, k, **kw)
    torch.cuda.synchronize()
    L.check(L.lib().a8_debug_set_stamps(None), "stamps")
    for kk in opt.get("tune", {}):
        L.check(L.lib().a8_set_tuning(kk, {14: 10000}.get(kk, 0)), "tune")
    st = stamps.view(-1, 8)
    nb = int((st[:, 4] != 0).sum())                                          # blocks that ran (tile shapes differ by variant)
    cyc = (st[:nb, 6] - st[:nb, 5]).double().cpu()                           # shader-clock cycles of the main loop
    st = st[:nb, :5].double().cpu() / 100.0                                  # us
    mhz = float((cyc / (st[:, 3] - st[:, 2]).clamp_min(1e-3)).median())
    t0 = st[:, 0].min()
    ph = [(st[:, i + 1] - st[:, i]) for i in range(4)]
    print("%-28s %5d blocks: kernel span %.1f us | block start spread %.1f us (first round) | per block, mean (p90) us: prologue->DMA issued %.2f (%.2f), "
          "first k-tile lands %.2f (%.2f), main loop %.2f (%.2f), epilogue %.2f (%.2f), total %.2f | shader clock over the main loop %.0f MHz" % (
              name, nb, float(st[:, 4].max() - t0), float(st[: min(nb, 512), 0].max() - t0),
              float(ph[0].mean()), float(ph[0].quantile(0.9)), float(ph[1].mean()), float(ph[1].quantile(0.9)),
              float(ph[2].mean()), float(ph[2].quantile(0.9)), float(ph[3].mean()), float(ph[3].quantile(0.9)), float((st[:, 4] - st[:, 0]).mean()), mhz), flush=True)
