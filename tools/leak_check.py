#!/usr/bin/env python3
"""200 bench steps: device memory (allocated / reserved) must be flat after warm-up."""
import os
import sys

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
marks = []
for i in range(200):
    st.zero_grad()
    loss = loss_fn(model, x)
    loss.backward()
    W.prefetch_host_draws(model, loss_fn, x.shape)
    v = loss.item()
    st.step()
    if i in (10, 50, 100, 199):
        torch.cuda.synchronize()
        marks.append((i, torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30, v))
for m in marks:
    print("step %d: allocated %.3f GiB reserved %.3f GiB loss %.4f" % m)
assert marks[-1][2] <= marks[1][2] * 1.05 + 0.1, "reserved memory keeps growing"
assert np.isfinite(marks[-1][3])
print("ok")
