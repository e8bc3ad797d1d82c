#!/usr/bin/env python3
"""Host time spent inside each autograd Function's backward / forward (Python side of audio8_amd.ops) per bench step."""
import collections
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
acc = collections.defaultdict(lambda: [0.0, 0])


def wrap(cls, name):
    fn = getattr(cls, name)

    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            e = acc[(cls.__name__, name)]
            e[0] += time.perf_counter() - t0
            e[1] += 1
    setattr(cls, name, staticmethod(timed))


for n in dir(ops):
    c = getattr(ops, n)
    if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function:
        wrap(c, "forward")
        wrap(c, "backward")


def step():
    st.zero_grad()
    loss = loss_fn(model, x)
    t0 = time.perf_counter()
    loss.backward()
    tb = time.perf_counter() - t0
    v = loss.item()
    st.step()
    return tb


for _ in range(3):
    step()
acc.clear()
N = 10
tb = sum(step() for _ in range(N))
print("backward() host time %.3f ms/step" % (tb / N * 1e3))
tot = {"forward": 0.0, "backward": 0.0}
for (c, n), (t, k) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print("%-28s %-8s %7.3f ms/step  %5.1f calls/step  %6.1f us/call" % (c, n, t / N * 1e3, k / N, t / k * 1e6))
    tot[n] += t / N * 1e3
print("sum inside Functions: forward %.3f ms, backward %.3f ms" % (tot["forward"], tot["backward"]))
