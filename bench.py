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
import glob
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver (set before HIP starts)

# Host placement, decided BEFORE torch is imported and before any GPU call (every thread the process starts later inherits it):
# each rank launches from its own small slice of the host's cores -- rank r of W takes --host-cores N (default 4) cores from the r-th
# of W equal slices of the allowed CPU list -- with as many OpenMP threads.  Unpinned, with one OpenMP thread per visible core, the
# launching thread of the SAME code is slower (host 9.5 vs 7.0 ms per step) and the step with it (11.86-11.92 vs 11.73-11.75 ms;
# 16 cores / 16 threads: 12.3 ms) -- profiles/r05_host_affinity.txt.  --host-cores 0 leaves the process unpinned; --host-cores 1 / 2 is
# the host-side stress of VERDICT r4 item 8b (the step must stay GPU-bound: host_ms_per_step < ms_per_step).
def _early_arg(name, default):
    """`--name V` or `--name=V` from argv, before argparse (the placement below must precede `import torch`)."""
    for i, a in enumerate(sys.argv):
        if a == name and i + 1 < len(sys.argv):
            return sys.argv[i + 1]
        if a.startswith(name + "="):
            return a.split("=", 1)[1]
    return default


# The launcher (`--gpus N` with no rendezvous in the environment) must NOT pin itself: its children inherit its mask, and N ranks
# slicing an already 4-core mask end up on the same four cores.  It hands the host's full CPU list to its ranks in
# A8_HOST_CPUS instead; a rank slices THAT list (or, started by an outside torchrun, the mask it inherited).
IS_LAUNCHER = int(_early_arg("--gpus", "1")) > 1 and "WORLD_SIZE" not in os.environ
ALL_CPUS = sorted(os.sched_getaffinity(0))
if os.environ.get("A8_HOST_CPUS") and not IS_LAUNCHER:
    try:
        ALL_CPUS = sorted(int(c) for c in os.environ["A8_HOST_CPUS"].split(",") if c != "") or ALL_CPUS
    except ValueError:
        pass
HOST_CORES = int(_early_arg("--host-cores", "0" if "--dry-run" in sys.argv else "4"))     # (a dry run pins only when asked to)


def host_slice(cpus, world, rank, cores):
    """The `cores` first CPUs of the rank's own 1/world slice of `cpus` (slices are disjoint; a slice shorter than `cores` is taken
    whole; more ranks than CPUs: ranks share round-robin)."""
    per = len(cpus) // max(1, world)
    if per == 0:
        return [cpus[rank % len(cpus)]]
    return cpus[rank * per:(rank + 1) * per][:cores]


if HOST_CORES > 0 and not IS_LAUNCHER:
    _w, _r = max(1, int(os.environ.get("WORLD_SIZE", "1"))), int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
    _mine = host_slice(ALL_CPUS, _w, _r % _w, HOST_CORES)
    try:
        os.sched_setaffinity(0, _mine)
        os.environ["OMP_NUM_THREADS"] = str(len(_mine))
        os.environ["MKL_NUM_THREADS"] = str(len(_mine))
    except OSError:                              # (a sandbox that forbids it: run unpinned)
        HOST_CORES = 0

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_TBS = 8.0             # HBM3E spec (same table; 6.3 TB/s is what a streaming copy reaches)
SAMPLE_RATE = 16000


def synth(B, L, rank):
    g = torch.Generator().manual_seed(1234 + rank)
    return (0.1 * torch.randn(B, L, generator=g)).clamp_(-1, 1)


def cpu_baseline(args):
    """The oracle (CPU restatement of the reference path, oracle/ref_cpu.py) driven through the same step
    sequence on this host's cores, on a bounded sample of the workload."""
    from oracle import ref_cpu as R
    B, L = args.cpu_batch, args.samples
    for tid in os.listdir("/proc/self/task"):                                # the CPU leg may use the whole CPU share again: the GPU legs were pinned,
        try:                                                                 # and worker threads started meanwhile inherited the pinned mask
            os.sched_setaffinity(int(tid), ALL_CPUS)
        except OSError:
            pass
    torch.set_num_threads(min(args.cpu_threads, len(ALL_CPUS)))              # the GPU box's CPU share for one GPU is 16
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
            "sample": f"wav2vec2-base fp32 dropout-free oracle step, batch {B} x {L} samples, 1 warm-up + {args.cpu_steps} timed "
                      f"(median {step_s:.2f} s, min {min(times):.2f} s per step)"}


def component_work(args, model, x):
    """Algorithmic work of the two north-star components for this workload (SURVEY 8d conventions), forward only:
    encoder layer = 8 B T d^2 + 4 B T d d_ff + 4 B T^2 d FLOP; conv layer i = 2 B Lout_i 512 Cin_i k_i FLOP;
    conv-stack bytes = 4 B L + 2 bpe sum_i(B 512 Lout_i) + bpe * weights."""
    B, L = x.shape
    specs = model.feature_extractor.conv_specs
    lens, cin, flop, elems, wts = [], 1, 0.0, 0, 0
    n = L
    for co, k, st in specs:
        n = (n - k) // st + 1
        lens.append(n)
        flop += 2.0 * B * n * co * cin * k
        elems += B * co * n
        wts += co * cin * k
        cin = co
    T = lens[-1]
    d = model.encoder.d_model
    layers = model.encoder.transformer.encoders
    dff = layers[0].ffn[0].layer.weight.shape[0] if hasattr(layers[0], "ffn") else 4 * d
    enc = len(layers) * (8.0 * B * T * d * d + 4.0 * B * T * d * dff + 4.0 * B * T * T * d)
    return {"enc_fwd_flop": enc, "conv_fwd_flop": flop, "conv0_out_elems": B * specs[0][0] * lens[0],
            "conv_fwd_bytes": lambda bpe: 4.0 * B * L + 2.0 * bpe * elems + bpe * wts}


def spawn_ranks(n):
    """`python bench.py --gpus N` with no rendezvous in the environment: become the launcher.  This process never
    touches the GPU; it starts N ranks (one per GPU) through torch.distributed.run with the same arguments, lets
    rank 0's JSON line through on stdout and returns the children's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # the ranks place themselves (prelude above) inside the launcher's UNPINNED mask, handed over explicitly
    return subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"),
                                         A8_HOST_CPUS=",".join(str(c) for c in ALL_CPUS)))


def dry_run(args):
    """The N-rank contract without a GPU: every rank joins a gloo group from the launcher's environment, exchanges a tensor, meets
    at a barrier; rank 0 prints ONE JSON line.  Rehearses `--gpus 8` (spawn, rendezvous on 127.0.0.1, exit codes) where eight
    processes may not share the one test GPU."""
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group(backend="gloo", init_method="env://", rank=rank, world_size=world)
    if os.environ.get("A8_BENCH_FAIL_AFTER_INIT") == str(rank):
        raise RuntimeError("bench.py: injected failure of rank %d after init_process_group" % rank)
    t = torch.tensor([float(rank + 1)])
    masks = [sorted(os.sched_getaffinity(0))]
    if world > 1:
        dist.all_reduce(t)
        masks = [None] * world
        dist.all_gather_object(masks, sorted(os.sched_getaffinity(0)))
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "dry run (launcher / rendezvous rehearsal, no GPU work)", "value": None, "n_gpus": world, "dry_run": True,
                          "rank_sum": float(t.item()), "config": {"global_batch": world * args.batch, "parallelism": "dp%d" % world,
                                                                  "host_cores": HOST_CORES, "rank_cpu_masks": masks}}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


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
    ap.add_argument("--cpu-batch", type=int, default=2)       # the reduced cfg2' [2, 250000], 1 warm-up + >= 2 timed
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--h2d-steps", type=int, default=10, help="extra steps fed from host memory through the on-ramp (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--bucket-mb", type=float, default=32.0)
    ap.add_argument("--dry-run", action="store_true", help="launcher / rendezvous rehearsal on CPU ranks (gloo): no GPU work, no metric")
    ap.add_argument("--host-cores", type=int, default=4, help="cores (and OpenMP threads) this rank launches from, taken from its slice of the host; 0 = unpinned")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    # fault injection for the launcher tests: the named rank dies before / after the rendezvous (never set in a measurement)
    if os.environ.get("A8_BENCH_FAIL_BEFORE_INIT") == os.environ.get("RANK", "0"):
        raise RuntimeError("bench.py: injected failure of rank %s before init_process_group" % os.environ.get("RANK", "0"))
    if args.dry_run:
        return dry_run(args)

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
    torch.manual_seed(rank)                    # every rank initialises on its own, as under DDP; the Stepper's C1 broadcast makes rank 0's win
    np.random.seed(1000 + rank)
    cfg = dict(dropout=args.dropout)
    if args.model == "large":
        cfg.update(d_model=1024, num_heads=16, num_layers=24, d_ff=4096)
    model = W.create_model(16, **cfg).to(dev).train()
    loss_fn = W.create_loss(640, 100)
    if os.environ.get("A8_HALF_BATCH", "1") == "0":                         # developer knob (tools/prof_serial.sh): one stream, whole batch
        ops.HALF_BATCH_STREAMS = False
    stepper = Stepper(model, lr=2.0e-4, weight_decay=1.0e-2, clip=1.0, bucket_mb=args.bucket_mb,
                      wgrad_side_stream=os.environ.get("A8_SIDE_WGRAD", "1") != "0")
    ops.GROUP_WGRAD = os.environ.get("A8_GROUP_WGRAD", "1") != "0"
    ops.ATTN_BIAS_FUSED = os.environ.get("A8_ATTN_BIAS_FUSED", "0") != "0"
    prefetch = os.environ.get("A8_PREFETCH_DRAWS", "1") != "0"
    if os.environ.get("A8_CONV_CHUNK"):                                      # developer knob: "utterances[,layers]"
        cc = [int(v) for v in os.environ["A8_CONV_CHUNK"].split(",")]
        ops.CONV_CHUNK_UTTS = cc[0]
        if len(cc) > 1:
            ops.CONV_CHUNK_LAYERS = cc[1]
    for kv in filter(None, os.environ.get("A8_TUNE", "").split(",")):       # developer knobs, e.g. A8_TUNE=5=0,1=403
        k, v = kv.split("=")
        L.check(L.lib().a8_set_tuning(int(k), int(v)), "a8_set_tuning")
    x = synth(args.batch, args.samples, rank).to(dev)

    host_s = {"fwd": 0.0, "bwd": 0.0, "draws": 0.0, "wait": 0.0, "opt": 0.0}      # host time per phase (enqueue cost; "wait" = loss.item())

    def one_step():
        c0 = time.perf_counter()
        stepper.zero_grad()
        loss = loss_fn(model, x)
        c1 = time.perf_counter()
        loss.backward()
        c2 = time.perf_counter()
        if prefetch:
            W.prefetch_host_draws(model, loss_fn, x.shape, dev)   # next step's numpy draws, while the GPU runs this backward
        c3 = time.perf_counter()
        lv = loss.item()                       # audio8/pretrain.py:180
        c4 = time.perf_counter()
        stepper.step()
        c5 = time.perf_counter()
        for k_, v_ in zip(("fwd", "bwd", "draws", "wait", "opt"), (c1 - c0, c2 - c1, c3 - c2, c4 - c3, c5 - c4)):
            host_s[k_] += v_
        return lv

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        lv = one_step()
    fence()
    for k_ in host_s:
        host_s[k_] = 0.0
    t0 = time.time()
    for _ in range(args.steps):
        lv = one_step()
    fence()
    dt = time.time() - t0
    host_ms = {k_: v_ / args.steps * 1e3 for k_, v_ in host_s.items()}
    if world > 1:
        tmax = torch.tensor([dt] + [host_ms[k_] for k_ in sorted(host_ms)], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax[0].item())
        host_ms = {k_: float(v_) for k_, v_ in zip(sorted(host_ms), tmax[1:].tolist())}
    ms_per_step = dt / args.steps * 1e3
    value = world * args.batch * args.samples / SAMPLE_RATE / (dt / args.steps)

    # ---- what audio8/pretrain.py:172-177 actually brackets: next(train_itr) hands over HOST arrays and batch.to(device) is inside
    # the timed step.  Same step, batch taken from host memory every step through the pinned double-buffered on-ramp
    # (audio8_amd.onramp: the upload of batch n+1 overlaps step n).  Reported next to `value`, never as `value`.
    ms_h2d = None
    if args.h2d_steps > 0:
        from audio8_amd.onramp import BatchOnRamp
        host = [row for row in x.cpu().numpy()]
        ramp = BatchOnRamp(depth=2)
        resident = x

        def h2d_step():
            nonlocal x
            x = ramp.get()
            ramp.submit(host)
            return one_step()
        ramp.submit(host)
        for _ in range(2):
            h2d_step()
        fence()
        t1 = time.time()
        for _ in range(args.h2d_steps):
            lv = h2d_step()
        fence()
        ms_h2d = (time.time() - t1) / args.h2d_steps * 1e3
        ramp.get()
        x = resident
        if world > 1:
            tmax = torch.tensor([ms_h2d], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            ms_h2d = float(tmax.item())

    # ---- live roofline: two extra instrumented steps after the timed region, HIP events on the launch stream around every
    # a8_gemm launch (dominant kernel) and around the conv-extractor / encoder launch sequences (the north-star components)
    roofline = None

    def instrumented():                                        # EVERY rank runs these steps (they contain the gradient collectives);
        K.GEMM_PROFILE = [] if rank == 0 else None             # only rank 0 brackets its launches with events
        ops.PHASE_EVENTS = {} if rank == 0 else None
        one_step()
        torch.cuda.synchronize()
        prof, K.GEMM_PROFILE = K.GEMM_PROFILE or [], None
        ph, ops.PHASE_EVENTS = ops.PHASE_EVENTS or {}, None
        phase_ms = {k: sum(a.elapsed_time(b) for a, b in v) for k, v in ph.items()}
        return prof, sum(p[0].elapsed_time(p[1]) for p in prof), sum(p[2] for p in prof), phase_ms
    prof, tot_ms, tot_fl, ph_run = instrumented()              # as the step runs (weight-gradient GEMMs overlap on a side stream)
    side, ops.WGRAD_SIDE_STREAM = ops.WGRAD_SIDE_STREAM, None
    # the shader clock the GEMM main loops actually ran at: every block of the serialised step stamps the 100 MHz counter and the
    # shader-clock counter at both ends of its main loop (a8_debug_set_stamps); under MFMA load the part runs well below 2.4 GHz
    # (a8_debug_set_stamps exists only in a developer build of the library, `make STAMPS=1`; the shipped one has no such export)
    have_stamps = hasattr(L.lib(), "a8_debug_set_stamps")
    stamps = torch.zeros(16384 * 8, device=dev, dtype=torch.int64) if (rank == 0 and have_stamps) else None
    if stamps is not None:
        L.check(L.lib().a8_debug_set_stamps(stamps.data_ptr()), "a8_debug_set_stamps")
    try:
        sprof, ser_ms, ser_fl, ph_ser = instrumented()         # the same launches one after another (each kernel's own rate)
    finally:
        if stamps is not None:
            L.check(L.lib().a8_debug_set_stamps(None), "a8_debug_set_stamps")
    ops.WGRAD_SIDE_STREAM = side
    clock_mhz = None
    if stamps is not None:
        st = stamps.view(-1, 8).cpu()
        ok = (st[:, 3] > st[:, 2]) & (st[:, 6] > st[:, 5])
        if int(ok.sum()) > 100:
            clock_mhz = float(((st[ok, 6] - st[ok, 5]).double() / ((st[ok, 3] - st[ok, 2]).double() / 100.0)).median())
    if rank == 0:
        if args.gemm_table:
            tab = {}
            for e0, e1, fl, lay, _, shp in sprof:
                t = tab.setdefault((lay,) + shp + (int(fl / 1e6),), [0, 0.0, fl])     # (grouped launches of different sizes share a shape tag)
                t[0] += 1
                t[1] += e0.elapsed_time(e1)
            for k, (n, ms, fl) in sorted(tab.items(), key=lambda kv: -kv[1][1]):
                print("gemm %s M=%d N=%d K=%d splitk=%d  n=%3d  avg %7.1f us  total %6.3f ms  %6.1f TFLOP/s"
                      % (("NT", "NN", "TN")[k[0]], k[1], k[2], k[3], k[4], n, ms / n * 1e3, ms, fl * n / ms / 1e9), file=sys.stderr)
            print("phases (serialised step, ms):", {k: round(v, 3) for k, v in sorted(ph_ser.items())}, file=sys.stderr)
        ach = tot_fl / (tot_ms * 1e-3) / 1e12
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3
        default_wl = args.dtype == "bf16" and args.model == "base" and args.batch == 8 and args.samples == 250000
        traffic, traffic_src, conv_pmc = None, None, None
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_gemm_traffic.json")))
        if cands and default_wl:
            tj = json.load(open(cands[-1]))        # HBM bytes per launch from separate rocprofv3 --pmc passes of this same command (newest round)
            # (per launch of THIS line's launch count: the timed step runs the encoder as two half-batches, i.e. more and smaller
            # launches than the instrumented whole-batch step whose algorithmic bytes stand beside it; the step total is what compares)
            traffic = round(tj["hbm_bytes_per_step"] / max(1, len(prof))) if "hbm_bytes_per_step" in tj else round(tj["hbm_bytes_per_launch"])
            traffic_src = "profiles/%s (%s; %.2f GB per step over the GEMM kernels of the step as timed)" % (
                os.path.basename(cands[-1]), tj["method"], tj.get("hbm_bytes_per_step", 0) / 1e9)
            conv_pmc = tj.get("conv_extractor_fwd_hbm_bytes")
        # ---- north-star components (SURVEY 8d algorithmic work; times from the SERIALISED instrumented step: every kernel of the
        # component back to back on one stream -- its own rate, no credit for overlap)
        comp = component_work(args, model, x)
        enc_ms = ph_ser.get("enc_fwd", 0.0) + ph_ser.get("enc_bwd", 0.0)
        conv_f, conv_b = ph_ser.get("conv_fwd", 0.0), ph_ser.get("conv_bwd", 0.0)
        c0_f, c0_b = ph_ser.get("conv0_fwd", 0.0), ph_ser.get("conv0_bwd", 0.0)
        esz = 2 if args.dtype == "bf16" else 4

        def frac(work, ms, pk):
            return round(work / (ms * 1e-3) / pk, 4) if ms > 0 else None
        encoder = {"gflop": round(3 * comp["enc_fwd_flop"] / 1e9, 1), "ms": round(enc_ms, 3), "ms_fwd": round(ph_ser.get("enc_fwd", 0.0), 3),
                   "ms_bwd": round(ph_ser.get("enc_bwd", 0.0), 3), "frac_mfma": frac(3 * comp["enc_fwd_flop"], enc_ms, peak * 1e12),
                   "ms_as_run": round(ph_run.get("enc_fwd", 0.0) + ph_run.get("enc_bwd", 0.0), 3),
                   "note": "12 x (QKV/O/FFN GEMMs + attention + 2 LayerNorms), forward + backward incl. weight gradients; algorithmic FLOPs 3 x forward"}
        conv_bytes_f = comp["conv_fwd_bytes"](esz)
        conv = {"algorithmic_bytes": round(3 * conv_bytes_f), "algorithmic_bytes_fwd": round(conv_bytes_f), "pmc_bytes_fwd": conv_pmc,
                "ms": round(conv_f + conv_b, 3), "ms_fwd": round(conv_f, 3), "ms_bwd": round(conv_b, 3),
                "frac_hbm": frac(3 * conv_bytes_f, conv_f + conv_b, PEAK_HBM_TBS * 1e12), "frac_hbm_fwd": frac(conv_bytes_f, conv_f, PEAK_HBM_TBS * 1e12),
                "gflop": round(3 * comp["conv_fwd_flop"] / 1e9, 1), "frac_mfma": frac(3 * comp["conv_fwd_flop"], conv_f + conv_b, peak * 1e12),
                "layer0": {"bytes_fwd": round(comp["conv0_out_elems"] * esz), "ms_fwd": round(c0_f, 3), "ms_bwd": round(c0_b, 3),
                           "frac_hbm_fwd": frac(comp["conv0_out_elems"] * esz, c0_f, PEAK_HBM_TBS * 1e12),
                           "frac_hbm_bwd": frac(comp["conv0_out_elems"] * esz, c0_b, PEAK_HBM_TBS * 1e12)},
                "note": "7-layer strided conv stack; bytes = SURVEY 8d minimum (input + each layer output written once and read once + weights), "
                        "backward counted as 2 x forward; layers 1-6 are MFMA-bound GEMMs (frac_mfma), only layer 0 is HBM-bound (layer0)"}
        roofline = {"bound": "mfma", "kernel": "a8_gemm kernel family <%s> (128 x 128-tile gemm_a3 / grouped / TN kernels + the BM x 256-tile gemm_stream kernel: all Linear/conv-as-GEMM fwd, dgrad, wgrad)" % args.dtype,
                    "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                    "achieved_serial": round(ser_fl / (ser_ms * 1e-3) / 1e12, 2), "frac_serial": round(ser_fl / (ser_ms * 1e-3) / 1e12 / peak, 4),
                    "traffic_unit": "HBM bytes per launch (PMC)", "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": round(sum(p[4] for p in prof) / max(1, len(prof))),
                    "launches_per_step": len(prof), "avg_launch_us": round(tot_ms * 1e3 / max(1, len(prof)), 2),
                    "gemm_ms_per_step": round(tot_ms, 3), "algorithmic_gflop_per_step": round(tot_fl / 1e9, 1),
                    "encoder": encoder, "conv_extractor": conv, "peak_hbm_TBs": PEAK_HBM_TBS,
                    "shader_clock_mhz_in_gemm": round(clock_mhz) if clock_mhz else None,
                    "frac_serial_at_measured_clock": round(ser_fl / (ser_ms * 1e-3) / 1e12 / (peak * clock_mhz / 2400.0), 4) if clock_mhz else None,
                    "clock_note": "peak is the guide's dense figure at the 2.4 GHz boost clock; the median shader clock over the GEMM main loops of the "
                                  "serialised step is what the power management sustains under this load"}
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
                          "last_loss": round(float(lv), 5),
                          "host_ms_per_step": {k_: round(v_, 3) for k_, v_ in host_ms.items()},     # max over ranks: enqueue time per phase ("wait" = loss.item())
                          "host_cores": HOST_CORES if HOST_CORES > 0 else len(ALL_CPUS),      # cores the GPU legs' launching process was confined to
                          "ms_per_step_with_h2d": round(ms_h2d, 3) if ms_h2d is not None else None,
                          "h2d_note": "same step with the batch handed over as host arrays every step (audio8/pretrain.py:172-177 times "
                                      "next(train_itr) + batch.to(device) too): pinned double-buffered on-ramp, %d steps" % args.h2d_steps},
               "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
