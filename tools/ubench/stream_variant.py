#!/usr/bin/env python3
"""One stream arrangement per PROCESS (HIP multiplexes streams over a few hardware queues: a variant that merely ADDS streams to the default set measures
queue sharing, not the variant -- tools/ubench/stream_prio_ab.py and cu_mask_ab.py did exactly that and saw 17-25 ms steps).  usage:
  stream_variant.py base | auxhi | side34 | side12 | extra2       -> one line: median / min ms per step over 4 x 10 steps
extra2: two MORE streams beside the step's four, each given one tiny kernel per step (what a communication library's own streams look like to the runtime);
run it with and without GPU_MAX_HW_QUEUES=8 in the environment (HIP's default is 4 hardware queues per device).
auxhi: the half-batch stream is created with priority -1;  sideNN: the weight-gradient stream is created with a CU mask keeping 3/4 (1/2) of every XCD's CUs."""
import ctypes as C
import os
import sys
import time

os.sched_setaffinity(0, sorted(os.sched_getaffinity(0))[:4])
os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from audio8_amd import ops, trainer, wav2vec2 as W  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "base"
dev = torch.device("cuda", 0)
torch.cuda.init()


def masked_stream(n):
    hip = C.CDLL("libamdhip64.so")
    words = (C.c_uint32 * 8)()
    for i in range(256):
        if (i // 8) % n != 0:
            words[i // 32] |= 1 << (i % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask -> %d" % rc)
    return torch.cuda.ExternalStream(s.value)


if variant == "auxhi":
    ops._HALF_AUX[dev] = torch.cuda.Stream(device=dev, priority=-1)
real_stream = torch.cuda.Stream
if variant in ("side34", "side12"):
    made = []

    def factory(*a, **k):                      # Stepper.__init__ creates the exchange stream first, the weight-gradient stream second
        made.append(1)
        return masked_stream(4 if variant == "side34" else 2) if len(made) == 2 else real_stream(*a, **k)
    trainer.torch.cuda.Stream = factory

ops.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
np.random.seed(0)
model = W.create_model(16).cuda().train()
loss_fn = W.create_loss(640, 100)
st = trainer.Stepper(model)
trainer.torch.cuda.Stream = real_stream
x = (0.1 * torch.randn(8, 250000, generator=torch.Generator().manual_seed(1234))).clamp_(-1, 1).cuda()


extra = [torch.cuda.Stream(), torch.cuda.Stream()] if variant == "extra2" else []
tiny = torch.zeros(64, device=dev)


def step():
    for e in extra:
        with torch.cuda.stream(e):
            tiny.add_(1.0)
    st.zero_grad()
    loss = loss_fn(model, x)
    loss.backward()
    W.prefetch_host_draws(model, loss_fn, x.shape)
    v = loss.item()
    st.step()
    return v


for _ in range(5):
    step()
ts = []
for _ in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 10 * 1e3)
print("%-7s median %.3f  min %.3f ms/step   (%s)" % (variant, float(np.median(ts)), min(ts), " ".join("%.3f" % t for t in ts)), flush=True)
st.close()
