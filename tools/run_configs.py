#!/usr/bin/env python3
"""Functional + timing check of BASELINE.json configs 4 (wav2vec2-large pretrain, 4 x 320k) and 5 (CTC fine-tune,
16 x 200k, 32 labels) at full size on one MI355X."""
import os
import sys
import time

os.sched_setaffinity(0, sorted(os.sched_getaffinity(0))[:4])       # launching thread placed as bench.py places it (profiles/r05_host_affinity.txt)
os.environ.setdefault("OMP_NUM_THREADS", "4")

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import ops, wav2vec2 as W, ctc as Cc  # noqa: E402
from audio8_amd.trainer import Stepper  # noqa: E402


def synth(B, L, seed=1234):
    g = torch.Generator().manual_seed(seed)
    return (0.1 * torch.randn(B, L, generator=g)).clamp_(-1, 1)


STEPS = 5


def timed(fn, warm=2, steps=None):
    steps = steps or STEPS
    for _ in range(warm):
        v = fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        v = fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / steps * 1e3, v


def cfg4():
    torch.manual_seed(0); np.random.seed(0)
    model = W.create_model(16, d_model=1024, num_heads=16, num_layers=24, d_ff=4096).cuda().train()
    loss_fn = W.create_loss(640, 100)
    st = Stepper(model)
    x = synth(4, 320000).cuda()

    def step():
        st.zero_grad(); loss = loss_fn(model, x); loss.backward(); W.prefetch_host_draws(model, loss_fn, x.shape); v = loss.item(); st.step(); return v
    ms, v = timed(step)
    print(f"cfg4 wav2vec2-large 4x320k bf16: {ms:.2f} ms/step, {4 * 320000 / 16000 / ms * 1e3:.0f} audio-s/s, loss {v:.4f}, params {sum(p.numel() for p in model.parameters()):,}")
    st.close()


def cfg5():
    torch.manual_seed(0); np.random.seed(0)
    Bn, L, V = 16, 200000, 32
    model = W.create_acoustic_model(V, 16).cuda().train()
    model.freeze = False                                  # encoder unfrozen, conv extractor frozen (freeze_fx default)
    st = Stepper(model, clip=25.0)
    loss_fn = Cc.CTCLoss(reduction_type="sum")
    x = synth(Bn, L).cuda()
    g = torch.Generator().manual_seed(1234)
    lengths = (torch.rand(Bn, generator=g) * 0.4 + 0.6) * L
    lengths = lengths.long(); lengths[0] = L
    pad_mask = (torch.arange(L).unsqueeze(0) < lengths.unsqueeze(1)).cuda()
    tl = torch.randint(50, 301, (Bn,), generator=g)
    targets = torch.full((Bn, 302), 1, dtype=torch.long)
    for b in range(Bn):
        targets[b, : tl[b]] = torch.randint(4, V, (int(tl[b]),), generator=g)
        targets[b, tl[b]] = 2
    targets = targets.cuda()

    def step():
        st.zero_grad()
        lp, pm = model(x, pad_mask)
        loss = loss_fn(lp.transpose(1, 0), pm.sum(-1), targets, tl)
        loss.backward(); v = loss.item(); st.step(); return v
    ms, v = timed(step)
    print(f"cfg5 CTC fine-tune 16x200k bf16: {ms:.2f} ms/step, {Bn * L / 16000 / ms * 1e3:.0f} audio-s/s, loss {v:.2f}")
    st.close()


if __name__ == "__main__":
    # usage: run_configs.py [cfg4|cfg5|both] [timed steps]   (under rocprofv3: `-- python3 tools/run_configs.py cfg5 3`)
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    if len(sys.argv) > 2:
        STEPS = int(sys.argv[2])
    ops.set_compute_dtype(torch.bfloat16)
    if os.environ.get("A8_HALF_BATCH", "1") == "0":        # developer A/B: the encoder stack as one whole-batch launch sequence
        ops.HALF_BATCH_STREAMS = False
    for kv in filter(None, os.environ.get("A8_TUNE", "").split(",")):       # developer knobs, e.g. A8_TUNE=19=1,5=1000
        from audio8_amd import _lib as L
        k, v = kv.split("=")
        L.check(L.lib().a8_set_tuning(int(k), int(v)), "a8_set_tuning")
    if which in ("cfg4", "both"):
        cfg4()
    if which in ("cfg5", "both"):
        cfg5()
