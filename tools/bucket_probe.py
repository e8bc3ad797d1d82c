#!/usr/bin/env python3
"""Which part of the gradient-bucket schedule costs step time on ONE GPU (round 6): the schedule off / complete / bookkeeping only (no GPU
work) / stream events only / stand-in kernels only / event RECORDS only (nobody waits) / events without the final join; interleaved rounds of
10 steps.  Result: the schedule's 0.35-0.45 ms per step for twelve buckets is the event record on the weight-gradient stream (0.30 ms by
itself); the Python bookkeeping and the stand-in kernels cost nothing, the waits and the final join 0.06 ms.  A follow-up with device-scope
events (hipEventReleaseToDevice) instead of torch's default ones measured no difference (11.69 vs 11.71 ms) and was not kept."""
import os, sys, time
os.sched_setaffinity(0, sorted(os.sched_getaffinity(0))[:4]); os.environ.setdefault("OMP_NUM_THREADS", "4")
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from audio8_amd import ops, wav2vec2 as W, _lib as L
from audio8_amd.trainer import Stepper
torch.manual_seed(0); np.random.seed(1000)
ops.set_compute_dtype(torch.bfloat16)
dev = torch.device("cuda", 0)
model = W.create_model(16, dropout=0.1).to(dev).train()
loss_fn = W.create_loss(640, 100)
st = Stepper(model, lr=2.0e-4, weight_decay=1.0e-2, clip=1.0)
red = st.reducer
x = (0.1 * torch.randn(8, 250000, generator=torch.Generator().manual_seed(1234))).clamp_(-1, 1).to(dev)
def step():
    st.zero_grad(); loss = loss_fn(model, x); loss.backward(); W.prefetch_host_draws(model, loss_fn, x.shape, dev); v = loss.item(); st.step(); return v
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
orig_launch = red._launch.__func__
def launch_none(self, bk): bk["launched"] = True
def launch_events_only(self, bk):
    for stt in dict(bk.get("streams", {})).values():
        ev = torch.cuda.Event(); ev.record(stt); self.side.wait_event(ev)
    bk["launched"] = True
def launch_kernel_only(self, bk):
    with torch.cuda.stream(self.side):
        L.check(L.lib().a8_delay_us(0.001, self.side.cuda_stream), "d")
    bk["launched"] = True
def launch_record_only(self, bk):
    for stt in dict(bk.get("streams", {})).values():
        ev = torch.cuda.Event(); ev.record(stt)
    bk["launched"] = True
JOIN = [True]
orig_wait = red.wait.__func__
def wait_maybe(self):
    if JOIN[0]:
        return orig_wait(self)
    self.handles = []
import types
red.wait = types.MethodType(wait_maybe, red)
variants = [("off", 0.0, None), ("full", 1e9, orig_launch), ("mark only", 1e9, launch_none), ("events only", 1e9, launch_events_only), ("kernel only", 1e9, launch_kernel_only),
            ("record only", 1e9, launch_record_only), ("events, no final join", 1e9, "nojoin")]
for _ in range(5): step()
res = {v[0]: [] for v in variants}
for r in range(4):
    for name, rate, fn in variants:
        red.fake_gbps = rate
        JOIN[0] = fn != "nojoin"
        if fn == "nojoin":
            fn = launch_events_only
        red._launch = types.MethodType(fn if fn else orig_launch, red)
        run(2); res[name].append(run(10))
for name, _, _ in variants:
    print("%-24s %.3f  %s" % (name, float(np.median(res[name])), " ".join("%.3f" % v for v in res[name])))
st.close()
