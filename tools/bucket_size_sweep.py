#!/usr/bin/env python3
"""Bucket size of the gradient exchange against its cost on ONE GPU (round 6): the bucket schedule itself costs ~30 us of step time per bucket
(tools/bucket_probe.py: the event record on the weight-gradient stream), fewer and larger buckets cost less -- as long as the last ones still
hide under the conv stack's backward.  For each bucket size: the step with zero-length stand-ins ("inf") and with delays of bytes / rate."""
import os
import sys
import time

os.sched_setaffinity(0, sorted(os.sched_getaffinity(0))[:4])
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import ops, wav2vec2 as W  # noqa: E402
from audio8_amd.trainer import Stepper  # noqa: E402

ops.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
np.random.seed(1000)
dev = torch.device("cuda", 0)
model = W.create_model(16, dropout=0.1).to(dev).train()
loss_fn = W.create_loss(640, 100)
x = (0.1 * torch.randn(8, 250000, generator=torch.Generator().manual_seed(1234))).clamp_(-1, 1).to(dev)
sizes = [float(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "32,64,96,128,192").split(",")]
rates = [0.0, 1e9, 90.0, 150.0, 300.0]
res = {}
for rnd in range(3):
    for mb in sizes:
        st = Stepper(model, lr=2.0e-4, weight_decay=1.0e-2, clip=1.0, bucket_mb=mb)

        def step():
            st.zero_grad(); loss = loss_fn(model, x); loss.backward(); W.prefetch_host_draws(model, loss_fn, x.shape, dev); v = loss.item(); st.step(); return v
        for _ in range(3):
            step()
        for rate in rates:
            st.reducer.fake_gbps = rate
            step(); step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            res.setdefault((mb, rate), []).append((time.perf_counter() - t0) / 10 * 1e3)
        nb = len(st.reducer.buckets)
        res[(mb, "n")] = nb
        st.reducer.fake_gbps = 0.0
        st.close()
print("# tools/bucket_size_sweep.py: wav2vec2-base 8 x 250k bf16, one MI355X, 3 interleaved rounds x 10 steps, ms/step (median)")
print("%-10s %8s %10s %10s %10s %10s %10s" % ("bucket MB", "buckets", "no sched", "inf", "90 GB/s", "150 GB/s", "300 GB/s"))
for mb in sizes:
    print("%-10g %8d " % (mb, res[(mb, "n")]) + " ".join("%10.3f" % float(np.median(res[(mb, r)])) for r in rates))
