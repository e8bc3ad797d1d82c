#!/usr/bin/env python3
"""How much of the gradient exchange is NOT hidden under backward -- measured on ONE GPU.

The data-parallel step issues one all-reduce per ~32 MB bucket of the flat gradient buffer on the exchange stream as backward completes
the bucket (audio8_amd/trainer.py GradReducer; the reference: DDP's reducer, audio8/pretrain.py:152-159,179).  Here every bucket's
collective is replaced by a one-wave delay kernel on that stream lasting bytes / rate (a8_delay_us): same buckets, same events, same
stream joins, no transport.  ms/step(rate) - ms/step(rate = inf) is the EXPOSED part of an exchange of that algorithm bandwidth
(bucket bytes / time: a ring over 8 xGMI-connected GPUs gives ~0.57 x the link rate, ~90 GB/s; the direct reduce-scatter + all-gather
over the 7 links ~4 x the link rate; RCCL lands in between), and t1 / (t1 + exposed) the weak-scaling efficiency it predicts at any rank count
with that per-bucket time.  What it cannot show: RCCL's kernels also occupy CUs and HBM bandwidth -- the delay kernel holds one wave.

usage: exchange_exposure.py [rounds] [steps per round]     (interleaved rounds: the boxes of the pool drift by a few %)"""
import os
import sys
import time

os.sched_setaffinity(0, sorted(os.sched_getaffinity(0))[:4])
os.environ.setdefault("OMP_NUM_THREADS", "4")

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import ops, wav2vec2 as W  # noqa: E402
from audio8_amd.trainer import Stepper  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    torch.manual_seed(0)
    np.random.seed(1000)
    ops.set_compute_dtype(torch.bfloat16)
    dev = torch.device("cuda", 0)
    model = W.create_model(16, dropout=0.1).to(dev).train()
    loss_fn = W.create_loss(640, 100)
    st = Stepper(model, lr=2.0e-4, weight_decay=1.0e-2, clip=1.0)
    red = st.reducer
    g = torch.Generator().manual_seed(1234)
    x = (0.1 * torch.randn(8, 250000, generator=g)).clamp_(-1, 1).to(dev)
    total_mb = st.state.total * 4 / 1e6

    host = {"bwd": 0.0, "opt": 0.0, "n": 0}

    def step():
        st.zero_grad()
        loss = loss_fn(model, x)
        t0 = time.perf_counter()
        loss.backward()
        t1 = time.perf_counter()
        W.prefetch_host_draws(model, loss_fn, x.shape, dev)
        v = loss.item()
        t2 = time.perf_counter()
        st.step()
        host["bwd"] += t1 - t0
        host["opt"] += time.perf_counter() - t2
        host["n"] += 1
        return v

    def run(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    variants = [("no bucket schedule", 0.0, "f32"), ("inf", 1e9, "f32")]
    for rate in (90.0, 150.0, 300.0, 600.0):
        for pl in ("f32", "bf16"):
            variants.append(("%g GB/s %s" % (rate, pl), rate, pl))
    for _ in range(5):
        step()
    res = {v[0]: [] for v in variants}
    hostt = {}
    for r in range(rounds):
        for name, rate, pl in variants:
            red.fake_gbps, red.payload = rate, pl
            run(2)
            host.update(bwd=0.0, opt=0.0, n=0)
            res[name].append(run(steps))
            hostt.setdefault(name, []).append((host["bwd"] / host["n"] * 1e3, host["opt"] / host["n"] * 1e3))
    red.fake_gbps, red.payload = 0.0, "f32"
    print("# tools/exchange_exposure.py: wav2vec2-base 8 x 250k bf16 on one MI355X, %d interleaved rounds x %d steps; gradient payload %.0f MB f32 in %d buckets"
          % (rounds, steps, total_mb, len(red.buckets)))
    base = float(np.median(res["inf"]))
    print("%-22s %9s %9s %9s   %s" % ("exchange stand-in", "ms/step", "exposed", "serial", "t1 / (t1 + exposed)"))
    for name, rate, pl in variants:
        ms = float(np.median(res[name]))
        serial = 0.0 if rate in (0.0, 1e9) else total_mb * (0.5 if pl == "bf16" else 1.0) / rate     # MB / (GB/s) = ms
        exposed = ms - base
        eff = "" if rate == 0.0 else "%.3f" % (base / max(base, ms))
        hb, ho = (float(np.median([h[k] for h in hostt[name]])) for k in (0, 1))
        print("%-22s %9.3f %9.3f %9.3f   %s   rounds: %s   host: backward %.2f ms, step() %.2f ms"
              % (name, ms, exposed, serial, eff, " ".join("%.3f" % v for v in res[name]), hb, ho))
    # when does each bucket go out, relative to the end of backward?  (one traced step at 300 GB/s, f32)
    red.fake_gbps, red.trace = 300.0, []
    st.zero_grad()
    t_start = torch.cuda.Event(enable_timing=True)
    t_start.record()
    loss = loss_fn(model, x)
    t_fwd = torch.cuda.Event(enable_timing=True)
    t_fwd.record()
    loss.backward()
    t_bwd = torch.cuda.Event(enable_timing=True)
    t_bwd.record()                                   # end of the data-gradient chain on the compute stream (weight gradients may still run on theirs)
    loss.item()
    st.step()
    torch.cuda.synchronize()
    tr, red.trace, red.fake_gbps = red.trace, None, 0.0
    print("# one traced step, 300 GB/s f32: forward %.2f ms, backward's last compute-stream launch at %.2f ms" % (t_start.elapsed_time(t_fwd), t_start.elapsed_time(t_bwd)))
    for b, e0, e1, nb in tr:
        print("  bucket %2d %6.1f MB  starts %7.3f ms  ends %7.3f ms  (%+.3f ms vs backward's end)"
              % (b, nb / 1e6, t_start.elapsed_time(e0), t_start.elapsed_time(e1), t_start.elapsed_time(e1) - t_start.elapsed_time(t_bwd)))
    st.close()


if __name__ == "__main__":
    main()
