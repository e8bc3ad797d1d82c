#!/usr/bin/env python3
"""Step time with the batch fed from host memory every step: pageable .to(device), vs the pinned double-buffered
on-ramp (audio8_amd.onramp.BatchOnRamp), vs a batch that is already resident (what bench.py times)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import ops, wav2vec2 as W  # noqa: E402
from audio8_amd.onramp import BatchOnRamp  # noqa: E402
from audio8_amd.trainer import Stepper  # noqa: E402

ops.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
np.random.seed(0)
model = W.create_model(16).cuda().train()
loss_fn = W.create_loss(640, 100)
st = Stepper(model)
B, Ls = 8, 250000
host = [np.clip(0.1 * np.random.randn(Ls + 1000 * i), -1, 1).astype(np.float32) for i in range(B)]     # ragged: cropped to 250000
resident = torch.from_numpy(np.stack([h[:Ls] for h in host])).cuda()
ramp = BatchOnRamp(depth=2)


def step(x):
    st.zero_grad()
    loss = loss_fn(model, x)
    loss.backward()
    W.prefetch_host_draws(model, loss_fn, x.shape)
    v = loss.item()
    st.step()
    return v


def run(mode, n=12):
    if mode == "onramp":
        ramp.submit(host)
    for i in range(n + 3):
        if i == 3:
            torch.cuda.synchronize()
            t0 = time.time()
        if mode == "resident":
            x = resident
        elif mode == "pageable":
            x = torch.from_numpy(np.stack([h[:Ls] for h in host])).to("cuda")
        else:
            x = ramp.get()
            ramp.submit(host)                            # next batch uploads while this step runs
        step(x)
    torch.cuda.synchronize()
    if mode == "onramp":
        ramp.get()
    return (time.time() - t0) / n * 1e3


for rnd in range(2):
    print("round", rnd, " ".join("%s %.2f ms/step" % (m, run(m)) for m in ("resident", "pageable", "onramp")))
