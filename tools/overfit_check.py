#!/usr/bin/env python3
"""End-to-end sanity of the bf16 training step at full size: repeat one batch (8 x 250 000 samples, fixed mask and
negatives via a fixed numpy seed per step) and watch the loss fall under AdamW.  Not a parity test: a smoke alarm for sign /
scaling mistakes anywhere in backward or the optimizer."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import ops, wav2vec2 as W  # noqa: E402
from audio8_amd.trainer import Stepper  # noqa: E402

ops.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
model = W.create_model(16, dropout=0.0).cuda().train()
loss_fn = W.create_loss(640, 100)
st = Stepper(model, lr=float(os.environ.get("LR", "5e-4")), weight_decay=0.0, clip=1.0)
x = (0.1 * torch.randn(8, 250000, generator=torch.Generator().manual_seed(1))).clamp_(-1, 1).cuda()
hist = []
for step in range(int(os.environ.get("STEPS", "60"))):
    np.random.seed(7)                         # same mask and negatives every step
    st.zero_grad()
    loss = loss_fn(model, x)
    loss.backward()
    hist.append(loss.item())
    st.step()
    if step % 10 == 0 or step == 59:
        print("step %3d  loss %.4f" % (step, hist[-1]), flush=True)
print("first 5 mean %.4f  last 5 mean %.4f" % (np.mean(hist[:5]), np.mean(hist[-5:])))
assert np.isfinite(hist).all() and np.mean(hist[-5:]) < np.mean(hist[:5]) - 0.2, "loss did not fall"
print("ok")
