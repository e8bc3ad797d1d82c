#!/usr/bin/env python3
"""cProfile of the host side of the bench step (GPU box)."""
import cProfile
import os
import pstats
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


def step():
    st.zero_grad()
    loss = loss_fn(model, x)
    loss.backward()
    v = loss.item()
    st.step()
    return v


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
torch.cuda.synchronize()
pr.disable()
ps = pstats.Stats(pr).sort_stats("tottime")
ps.print_stats(28)
