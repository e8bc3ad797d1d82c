#!/usr/bin/env python3
"""Which ATen kernels still run inside a bench step, and from which line of the package (torch profiler, with stacks)."""
import collections
import os
import sys

import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import ops, wav2vec2 as W  # noqa: E402
from audio8_amd.trainer import Stepper  # noqa: E402

ops.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
np.random.seed(0)
model = W.create_model(16).cuda().train()
loss_fn = W.create_loss(640, 100)
st = Stepper(model)
x = (0.1 * torch.randn(8, 250000, generator=torch.Generator().manual_seed(1234))).clamp_(-1, 1).cuda()


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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::"):
        continue
    if not getattr(ev, "kernels", []):
        continue
    frames = [f for f in (ev.stack or []) if "audio8_amd" in f or "bench" in f or "tools/" in f]
    where = frames[0] if frames else (ev.stack[0] if ev.stack else "?")
    cnt[(ev.name, str(getattr(ev, "input_shapes", ""))[:80] + " " + where.strip()[-110:])] += 1
for (name, where), n in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print("%3d  %-28s %s" % (n, name, where))
