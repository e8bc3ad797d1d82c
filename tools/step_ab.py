#!/usr/bin/env python3
"""Same-process, interleaved A/B of the full bench step under developer knobs (box-to-box and run-to-run noise on this pool is
several per cent; only interleaved rounds in one process separate 1 % effects).
usage: step_ab.py "name:key=val,key=val;name2:..."   keys: integers = a8_set_tuning keys; bucketnorm, sidewgrad, groupwgrad ... = 0/1; fakegbps = <rate>"""
import os
import sys
import time

# the launching thread on 4 cores with 4 OpenMP threads, as bench.py places it (profiles/r05_host_affinity.txt: unpinned, the host side
# jitters by several tenths of a millisecond per step -- more than most effects this tool is used to separate)
os.sched_setaffinity(0, sorted(os.sched_getaffinity(0))[:4])
os.environ.setdefault("OMP_NUM_THREADS", "4")

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import _lib as L, ops, wav2vec2 as W  # noqa: E402
from audio8_amd.trainer import Stepper  # noqa: E402

DEFAULTS = {1: 802, 2: 0, 3: 41, 4: 41, 5: 20000, 6: 3, 7: 0, 8: 0, 14: 1000, 15: 1, 17: 801, 19: 0, 20: 0, 21: 0, 22: 0, 23: 0, 25: 0}      # (key 24 selected the round-5 wide-tile kernel, which lives only in tools/ubench/round5_gemm_wide: no library code reads it)
cfgs = []
for part in (sys.argv[1] if len(sys.argv) > 1 else "base:").split(";"):
    name, _, kv = part.partition(":")
    d = {}
    for item in filter(None, kv.split(",")):
        k, v = item.split("=")
        d[int(k) if k.isdigit() else k] = int(v)
    cfgs.append((name, d))
ops.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
np.random.seed(0)
model = W.create_model(16).cuda().train()
loss_fn = W.create_loss(640, 100)
st = Stepper(model)
x = (0.1 * torch.randn(8, 250000, generator=torch.Generator().manual_seed(1234))).clamp_(-1, 1).cuda()
side = st.wgrad_stream


def apply(d):
    for k, v in DEFAULTS.items():
        L.check(L.lib().a8_set_tuning(k, d.get(k, v)), "tune")
    st.reducer.norm_sq = st.gnorm_sq if d.get("bucketnorm", 0) else None
    ops.WGRAD_SIDE_STREAM = side if d.get("sidewgrad", 1) else None
    ops.GROUP_WGRAD = bool(d.get("groupwgrad", 1))
    ops.CONV_CHUNK_UTTS = d.get("convchunk", 0)
    ops.SIDE_EXTRAS = bool(d.get("sidex", 1))
    ops.CONV_DGRAD_GROUPED = bool(d.get("dgroup", 1))
    ops.CONV_WGRAD_DEFER = d.get("wdefer", 0)
    ops.WPREP_SPLIT = bool(d.get("wsplit", 1))
    ops.COMPOSITE_ENCODER_FWD = bool(d.get("compfwd", 1))
    ops.COMPOSITE_ENCODER_BWD = bool(d.get("compbwd", 1))
    ops.WGRAD_STORE_FIRST = bool(d.get("storefirst", 1))
    ops.HALF_BATCH_STREAMS = bool(d.get("half", 1))
    ops.HALF_BATCH_BWD = bool(d.get("halfbwd", 1))
    ops.LN_PARTIALS = bool(d.get("lnp", 1))
    ops.HOST_DELAY_US = d.get("hostdelay", 0)
    ops.SPLITK_SLOTS = d.get("skslots", 512)
    ops.SPLITK_MIN_ROWS = d.get("skrows", 800)
    st.reducer.fake_gbps = float(d.get("fakegbps", 0))                # > 0: the bucket schedule with a delay standing in for every collective


def step():
    st.zero_grad()
    loss = loss_fn(model, x)
    loss.backward()
    W.prefetch_host_draws(model, loss_fn, x.shape)
    v = loss.item()
    st.step()
    return v


res = {n: [] for n, _ in cfgs}
nsteps = int(os.environ.get("AB_STEPS", "8"))
for rnd in range(int(os.environ.get("AB_ROUNDS", "5"))):
    for name, d in cfgs:
        apply(d)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step()
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / nsteps * 1e3)
    print("round %d: " % rnd + "  ".join("%s %.3f" % (n, res[n][-1]) for n, _ in cfgs), flush=True)
print("median ms/step: " + "  ".join("%s %.3f (min %.3f)" % (n, float(np.median(res[n])), min(res[n])) for n, _ in cfgs))
apply({})
