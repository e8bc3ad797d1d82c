#!/usr/bin/env python3
"""Where a bench step's wall time goes: host enqueue time vs GPU time of forward / backward / optimizer."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import ops, wav2vec2 as W  # noqa: E402
from audio8_amd.trainer import Stepper  # noqa: E402

ops.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
np.random.seed(0)
model = W.create_model(16).cuda().train()
loss_fn = W.create_loss(640, 100)
st = Stepper(model)
x = (0.1 * torch.randn(8, 250000)).clamp_(-1, 1).cuda()


def step(rec=None):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    t = [time.perf_counter()]
    ev[0].record()
    st.zero_grad()
    loss = loss_fn(model, x)
    ev[1].record(); t.append(time.perf_counter())
    loss.backward()
    ev[2].record(); t.append(time.perf_counter())
    if os.environ.get("A8_PREFETCH_DRAWS", "1") != "0":
        W.prefetch_host_draws(model, loss_fn, x.shape)
    v = loss.item()
    ev[3].record(); t.append(time.perf_counter())
    st.step()
    ev[4].record(); t.append(time.perf_counter())
    if rec is not None:
        rec.append((ev, t))
    return v


for _ in range(5):
    step()
torch.cuda.synchronize()
rec = []
t0 = time.perf_counter()
for _ in range(10):
    step(rec)
torch.cuda.synchronize()
t1 = time.perf_counter()
print("ms/step %.3f" % ((t1 - t0) * 100))
names = ["zero_grad+forward+loss", "backward", "loss.item() wait", "optimizer enqueue"]
for i, n in enumerate(names):
    host = np.mean([(t[i + 1] - t[i]) * 1e3 for _, t in rec])
    gpu = np.mean([ev[i].elapsed_time(ev[i + 1]) for ev, _ in rec])
    print("%-26s host %.3f ms   gpu-span %.3f ms" % (n, host, gpu))
# GPU span across steps: end-of-step event to next step's first event
gaps = [rec[i][0][4].elapsed_time(rec[i + 1][0][0]) for i in range(len(rec) - 1)]
print("gpu span between steps (optimizer kernels etc.): %.3f ms" % np.mean(gaps))
