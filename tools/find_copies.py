#!/usr/bin/env python3
"""Which Python lines issue the torch copy_/fill_/zero_ kernels of a bench step (torch.profiler with stacks)."""
import collections
import os
import sys

import numpy as np
import torch
from torch.profiler import profile, ProfilerActivity

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
    W.prefetch_host_draws(model, loss_fn, x.shape)
    v = loss.item()
    st.step()
    return v


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::zeros", "aten::add_", "aten::add", "aten::mul", "aten::neg", "aten::sum", "aten::div"):
        frames = [f for f in (e.stack or []) if "/root/repo" in f or "audio8_amd" in f or "bench" in f or "tools/" in f]
        agg[(e.name, frames[0] if frames else (e.stack[0] if e.stack else "?"))] += 1
for (name, where), n in sorted(agg.items(), key=lambda kv: -kv[1])[:60]:
    print("%3d  %-18s %s" % (n, name, where))
