#!/usr/bin/env python3
"""bench.py -- audio-seconds/second of one wav2vec2-base pretrain step (BASELINE.json metric).

A step is exactly what audio8/pretrain.py:171-186 times: zero_grad, forward, contrastive + diversity
loss, backward, loss.item(), clip_grad_norm_(1.0), AdamW.  Workload (config[1]): wav2vec2-base,
batch 8 x 250 000 samples of synthetic 16 kHz audio per GPU, bf16 activations (f32 accumulate /
statistics / master weights), dropout 0.1 as the reference's CLI defaults.  Inputs are resident in
HBM before the timed region.  N > 1: one process per GPU (torch.distributed.run), utterances sharded
across ranks (weak scaling), gradients exchanged by bucketed RCCL all-reduce overlapped with backward.

Prints ONE JSON line (rank 0) with the extra `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver (set before HIP starts)

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md, chip-level parameters)
SAMPLE_RATE = 16000


def synth(B, L, rank):
    g = torch.Generator().manual_seed(1234 + rank)
    return (0.1 * torch.randn(B, L, generator=g)).clamp_(-1, 1)


def cpu_baseline(args):
    """The oracle (CPU restatement of the reference path, oracle/ref_cpu.py) driven through the same step
    sequence on this host's cores, on a bounded sample of the workload."""
    from oracle import ref_cpu as R
    B, L = args.cpu_batch, args.samples
    torch.set_num_threads(min(args.cpu_threads, os.cpu_count() or 1))       # the GPU box's CPU share for one GPU is 16
    torch.manual_seed(0)
    np.random.seed(0)
    sd = {k: v.clone().requires_grad_(True) for k, v in R.init_state_dict(seed=0).items()}
    opt = torch.optim.AdamW(list(sd.values()), lr=2e-4, weight_decay=1e-2)
    x = synth(B, L, 0)
    times = []
    for it in range(1 + args.cpu_steps):
        t0 = time.time()
        opt.zero_grad()
        loss, _ = R.pretrain_loss(x, sd, training=True)
        loss.backward()
        loss.item()
        torch.nn.utils.clip_grad_norm_(list(sd.values()), 1.0)
        opt.step()
        if it > 0:
            times.append(time.time() - t0)
    step_s = float(np.median(times))
    return {"value": B * L / SAMPLE_RATE / step_s, "unit": "audio-sec/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"wav2vec2-base fp32 dropout-free oracle step, batch {B} x {L} samples, 1 warm-up + {args.cpu_steps} timed"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--samples", type=int, default=250000)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--model", default="base", choices=["base", "large"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gemm-table", action="store_true", help="print the per-shape table of the serialised instrumented step to stderr")
    ap.add_argument("--cpu-batch", type=int, default=1)
    ap.add_argument("--cpu-steps", type=int, default=1)
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--bucket-mb", type=float, default=32.0)
    args = ap.parse_args()

    from audio8_amd import _lib as L
    from audio8_amd import ops, wav2vec2 as W, kernels as K
    from audio8_amd.trainer import Stepper, init_distributed_from_env
    import torch.distributed as dist

    rank, local, world = init_distributed_from_env()
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a GPU: the hot path has no CPU fallback")
    dev = torch.device("cuda", torch.cuda.current_device())
    ops.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    torch.manual_seed(0)
    np.random.seed(1000 + rank)
    cfg = dict(dropout=args.dropout)
    if args.model == "large":
        cfg.update(d_model=1024, num_heads=16, num_layers=24, d_ff=4096)
    model = W.create_model(16, **cfg).to(dev).train()
    loss_fn = W.create_loss(640, 100)
    stepper = Stepper(model, lr=2.0e-4, weight_decay=1.0e-2, clip=1.0, bucket_mb=args.bucket_mb,
                      wgrad_side_stream=os.environ.get("A8_SIDE_WGRAD", "1") != "0")
    ops.GROUP_WGRAD = os.environ.get("A8_GROUP_WGRAD", "1") != "0"
    ops.ATTN_BIAS_FUSED = os.environ.get("A8_ATTN_BIAS_FUSED", "0") != "0"
    prefetch = os.environ.get("A8_PREFETCH_DRAWS", "1") != "0"
    for kv in filter(None, os.environ.get("A8_TUNE", "").split(",")):       # developer knobs, e.g. A8_TUNE=5=0,1=403
        k, v = kv.split("=")
        L.check(L.lib().a8_set_tuning(int(k), int(v)), "a8_set_tuning")
    x = synth(args.batch, args.samples, rank).to(dev)

    def one_step():
        stepper.zero_grad()
        loss = loss_fn(model, x)
        loss.backward()
        if prefetch:
            W.prefetch_host_draws(model, loss_fn, x.shape, dev)   # next step's numpy draws, while the GPU runs this backward
        lv = loss.item()                       # audio8/pretrain.py:180
        stepper.step()
        return lv

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        lv = one_step()
    fence()
    t0 = time.time()
    for _ in range(args.steps):
        lv = one_step()
    fence()
    dt = time.time() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * args.batch * args.samples / SAMPLE_RATE / (dt / args.steps)

    # ---- live roofline of the dominant kernel (a8_gemm, bf16 MFMA): one extra instrumented step,
    # HIP events on the launch stream around every a8_gemm launch
    roofline = None

    def gemm_rate():                                           # EVERY rank runs these steps (they contain the gradient collectives);
        K.GEMM_PROFILE = [] if rank == 0 else None             # only rank 0 brackets its launches with events
        one_step()
        torch.cuda.synchronize()
        prof, K.GEMM_PROFILE = K.GEMM_PROFILE or [], None
        return prof, sum(p[0].elapsed_time(p[1]) for p in prof), sum(p[2] for p in prof)
    prof, tot_ms, tot_fl = gemm_rate()                         # as the step runs (weight-gradient GEMMs overlap on a side stream)
    side, ops.WGRAD_SIDE_STREAM = ops.WGRAD_SIDE_STREAM, None
    sprof, ser_ms, ser_fl = gemm_rate()                        # the same launches one after another (kernel's own rate)
    ops.WGRAD_SIDE_STREAM = side
    if rank == 0:
        if args.gemm_table:
            tab = {}
            for e0, e1, fl, lay, _, shp in sprof:
                t = tab.setdefault((lay,) + shp, [0, 0.0, fl])
                t[0] += 1
                t[1] += e0.elapsed_time(e1)
            for k, (n, ms, fl) in sorted(tab.items(), key=lambda kv: -kv[1][1]):
                print("gemm %s M=%d N=%d K=%d splitk=%d  n=%3d  avg %7.1f us  total %6.3f ms  %6.1f TFLOP/s"
                      % (("NT", "NN", "TN")[k[0]], k[1], k[2], k[3], k[4], n, ms / n * 1e3, ms, fl * n / ms / 1e9), file=sys.stderr)
        ach = tot_fl / (tot_ms * 1e-3) / 1e12
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "r01_gemm_traffic.json")
        if os.path.exists(tpath) and args.dtype == "bf16" and args.model == "base" and args.batch == 8 and args.samples == 250000:
            tj = json.load(open(tpath))            # HBM bytes per launch from separate rocprofv3 --pmc passes of this same command
            traffic, traffic_src = round(tj["hbm_bytes_per_launch"]), "profiles/r01_gemm_traffic.json (" + tj["method"] + ")"
        roofline = {"bound": "mfma", "kernel": "gemm_kernel<%s> (a8_gemm: all Linear/conv-as-GEMM fwd, dgrad, wgrad)" % args.dtype,
                    "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                    "achieved_serial": round(ser_fl / (ser_ms * 1e-3) / 1e12, 2), "frac_serial": round(ser_fl / (ser_ms * 1e-3) / 1e12 / peak, 4),
                    "traffic_unit": "HBM bytes per launch (PMC)", "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": round(sum(p[4] for p in prof) / max(1, len(prof))),
                    "launches_per_step": len(prof), "avg_launch_us": round(tot_ms * 1e3 / max(1, len(prof)), 2),
                    "gemm_ms_per_step": round(tot_ms, 3), "algorithmic_gflop_per_step": round(tot_fl / 1e9, 1)}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)
    if rank == 0:
        out = {"metric": "audio-sec/s (pretrain step) wav2vec2-%s bs%dx%dk" % (args.model, args.batch, args.samples // 1000),
               "value": round(value, 2), "unit": "audio-sec/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": args.dtype, "data": "synthetic 16 kHz waveforms (0.1*randn clamped to [-1,1]), random-init weights",
               "config": {"workload": "wav2vec2-%s pretrain step (fwd + contrastive/diversity loss + bwd + clip + AdamW), batch %d x %d samples per GPU, dropout %.2f"
                                      % (args.model, args.batch, args.samples, args.dropout),
                          "global_batch": world * args.batch, "samples_per_clip": args.samples, "parallelism": "dp%d" % world,
                          "last_loss": round(float(lv), 5)},
               "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
