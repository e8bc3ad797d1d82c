#!/usr/bin/env python3
"""Stream priorities for the three-stream step (round 6): the two data-gradient chains (caller's stream + half-batch stream) wait 80-160 us for CU slots in
front of every LayerNorm backward (profiles/r06_bench_summary.txt: 2.4 ms of main-queue gaps per step) while the weight-gradient stream's grouped launches
hold the slots.  Variants, interleaved in one process:  base (all default priority) | auxhi (half-batch stream high) | hi (step on a high-priority stream, half-batch stream high) |
lowside (weight-gradient stream at the lowest priority the device offers) | both."""
import contextlib
import os
import sys
import time

os.sched_setaffinity(0, sorted(os.sched_getaffinity(0))[:4])
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from audio8_amd import ops, wav2vec2 as W  # noqa: E402
from audio8_amd.trainer import Stepper  # noqa: E402

ops.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
np.random.seed(0)
dev = torch.device("cuda", 0)
model = W.create_model(16).cuda().train()
loss_fn = W.create_loss(640, 100)
st = Stepper(model)
x = (0.1 * torch.randn(8, 250000, generator=torch.Generator().manual_seed(1234))).clamp_(-1, 1).cuda()
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("stream priority range (least, greatest):", lo, hi)
norm_aux = ops.half_stream(dev)
hi_main, hi_aux = torch.cuda.Stream(priority=hi), torch.cuda.Stream(priority=hi)
norm_side = st.wgrad_stream
low_side = torch.cuda.Stream(priority=lo)


def set_side(s):
    st.wgrad_stream = s
    ops.WGRAD_SIDE_STREAM = s
    st.wcache.side = s
    st.reducer.extra_streams[0] = s
    st.reducer._wgrad_raw = s.cuda_stream


def step():
    st.zero_grad()
    loss = loss_fn(model, x)
    loss.backward()
    W.prefetch_host_draws(model, loss_fn, x.shape)
    v = loss.item()
    st.step()
    return v


variants = {"base": (False, False, False), "auxhi": (False, False, True), "hi": (True, False, True), "lowside": (False, True, False), "both": (True, True, True)}
if lo == 0:
    variants = {k: v for k, v in variants.items() if not v[1]}          # (this device offers no priority below the default)
res = {k: [] for k in variants}
for rnd in range(5):
    for name, (hi_chain, low, aux_hi) in variants.items():
        torch.cuda.synchronize()
        ops._HALF_AUX[dev] = hi_aux if aux_hi else norm_aux
        set_side(low_side if low else norm_side)
        with (torch.cuda.stream(hi_main) if hi_chain else contextlib.nullcontext()):
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            res[name].append((time.perf_counter() - t0) / 10 * 1e3)
    print("round %d: " % rnd + "  ".join("%s %.3f" % (k, v[-1]) for k, v in res.items()), flush=True)
print("median ms/step: " + "  ".join("%s %.3f (min %.3f)" % (k, float(np.median(v)), min(v)) for k, v in res.items()))
ops._HALF_AUX[dev] = norm_aux
set_side(norm_side)
st.close()
