"""Data-parallel gradient exchange on CPU with gloo, world_size 2: the bucketed reducer (audio8_amd.trainer)
must leave every rank with the SUM of the per-rank gradients, including buckets whose parameters got no
gradient this step (frozen encoder case, audio8/train.py:266-268)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from audio8_amd.trainer import FlatState, GradReducer
        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Linear(40, 300), torch.nn.GELU(), torch.nn.Linear(300, 300), torch.nn.GELU(),
                                    torch.nn.Linear(300, 7))
        frozen_probe = torch.nn.Parameter(torch.zeros(1000))       # never used in forward: its bucket gets no hook
        model.register_parameter("unused", frozen_probe)
        st = FlatState(model)
        red = GradReducer(st, bucket_mb=0.2, tail_elems=2000)
        assert len(red.buckets) >= 2
        cover = sorted((b["lo"], b["hi"]) for b in red.buckets)          # the buckets partition the flat buffer
        assert cover[0][0] == 0 and cover[-1][1] == st.total and all(cover[i][1] == cover[i + 1][0] for i in range(len(cover) - 1))
        assert sorted(i for b in red.buckets for i in b["members"]) == list(range(len(st.params)))
        tail = red.buckets[-1]                                           # last-arriving gradients: a small bucket of their own
        assert tail["lo"] == 0 and 0 < tail["hi"] - tail["lo"] <= 2000 and len(red.buckets) >= 3
        ok = True
        for step in range(2):
            st.zero_grad()
            red.begin_step()
            g = torch.Generator().manual_seed(100 * step + rank)
            x = torch.randn(16, 40, generator=g)
            model(x).square().sum().backward()
            local = st.flat_g.clone() if False else None
            red.finish()
            # reference: recompute both ranks' gradients locally
            ref = torch.zeros_like(st.flat_g)
            for r in range(world):
                m2 = torch.nn.Sequential(torch.nn.Linear(40, 300), torch.nn.GELU(), torch.nn.Linear(300, 300), torch.nn.GELU(),
                                         torch.nn.Linear(300, 7))
                m2.load_state_dict({k: v for k, v in model.state_dict().items() if k != "unused"})
                xr = torch.randn(16, 40, generator=torch.Generator().manual_seed(100 * step + r))
                m2(xr).square().sum().backward()
                off = 0
                for p, o in zip(st.params, st.offsets):
                    pass
                for (n, p2) in m2.named_parameters():
                    i = [id(pp) for pp in st.params].index(id(dict(model.named_parameters())[n]))
                    ref[st.offsets[i]:st.offsets[i] + p2.numel()] += p2.grad.reshape(-1)
            ok = ok and torch.allclose(st.flat_g, ref, rtol=1e-5, atol=1e-6)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _bf16_payload_worker(rank, world, port, q):
    """payload="bf16": every bucket travels as bf16 (half the bytes on the links) and lands back in the f32 flat buffer.  Every rank must
    end with the SAME values (bit for bit: they all cast back the same reduced bf16 numbers), equal to the bf16-rounded sum of the
    bf16-rounded per-rank gradients, hence within 2^-6 of the f32 exchange relative to the larger addend (three roundings of relative
    2^-8 each, the sum at most twice the larger addend); untouched buckets stay zero."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from audio8_amd.trainer import FlatState, GradReducer
        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Linear(40, 300), torch.nn.GELU(), torch.nn.Linear(300, 300), torch.nn.GELU(), torch.nn.Linear(300, 7))
        model.register_parameter("unused", torch.nn.Parameter(torch.zeros(1000)))
        st = FlatState(model)
        red = GradReducer(st, bucket_mb=0.2, tail_elems=2000, payload="bf16")
        assert red.payload == "bf16" and len(red.buckets) >= 3
        ok = True
        for step in range(2):
            st.zero_grad()
            red.begin_step()
            x = torch.randn(16, 40, generator=torch.Generator().manual_seed(100 * step + rank))
            model(x).square().sum().backward()
            local = st.flat_g.clone()
            red.finish()
            both = [torch.empty_like(local) for _ in range(world)]
            dist.all_gather(both, local)                                   # the per-rank f32 gradients, for the reference
            exact = both[0] + both[1]
            want = (both[0].bfloat16() + both[1].bfloat16()).float()       # bf16 addends, one bf16-rounded sum (world 2: a single add)
            ok = ok and torch.equal(st.flat_g, want)
            scale = torch.maximum(both[0].abs(), both[1].abs())
            ok = ok and bool(((st.flat_g - exact).abs() <= 2.0 ** -6 * scale + 1e-30).all())
            same = [torch.empty_like(local) for _ in range(world)]
            dist.all_gather(same, st.flat_g)
            ok = ok and torch.equal(same[0], same[1])                      # replicas cannot drift: identical bits on every rank
            unused = [i for i, p in enumerate(st.params) if p is model.unused][0]
            ok = ok and bool((st.flat_g[st.offsets[unused]:st.offsets[unused] + 1000] == 0).all())
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_bf16_gradient_payload_gloo_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bf16_payload_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_bucketed_reducer_gloo_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def _w2v_worker(rank, world, port, q):
    """The wav2vec2 module itself (2 encoder layers) under the Stepper's data-parallel contract, CPU tensors + gloo:
    C1 broadcast, direct-gradient marks + autograd hooks, no_sync accumulation, one-collective batch counters,
    per-parameter has-gradient tracking.  Gradients come from the oracle's autograd (the HIP kernels need a GPU)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np
        from audio8_amd import wav2vec2 as W
        from audio8_amd.trainer import Stepper
        from oracle import ref_cpu as R
        torch.manual_seed(100 + rank)                       # every rank starts from its OWN initialisation (as under DDP)
        model = W.create_model(num_layers=2, dropout=0.0, dropout_input=0.0, dropout_features=0.0)
        own = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()
        st = Stepper(model, bucket_mb=4.0)                  # CPU: FlatState + reducer (+ broadcast); no kernels are launched
        red, fs = st.reducer, st.state
        assert red.enabled and len(red.buckets) >= 3
        # C1: rank 0's parameters everywhere, and the module's parameters ARE the flat views
        gathered = [torch.empty_like(fs.flat_p) for _ in range(world)]
        dist.all_gather(gathered, fs.flat_p)
        ok = all(torch.equal(g, gathered[0]) for g in gathered)
        if rank != 0:
            ok = ok and not torch.equal(torch.cat([p.detach().reshape(-1) for p in model.parameters()]), own)
        sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
        names = {id(p): n for n, p in model.named_parameters()}

        def oracle_grads(seed):
            np.random.seed(seed)
            x = (0.1 * torch.randn(2, 16000, generator=torch.Generator().manual_seed(seed))).clamp_(-1, 1)
            for v in sd.values():
                v.grad = None
            loss, _ = R.pretrain_loss(x, sd, num_layers=2, training=False)
            loss.backward()
            return {k: v.grad.clone() for k, v in sd.items() if v.grad is not None}

        def run_backward(grads, skip=()):
            """Emulates the backward pass: last parameter first; even ones as direct writes + marks, odd ones through autograd."""
            for i in range(len(fs.params) - 1, -1, -1):
                p = fs.params[i]
                n = names[id(p)]
                if any(n.startswith(s) for s in skip):
                    continue
                g = grads[n]
                if i % 2 == 0:
                    p.grad.add_(g)
                    red.param_ready(p)
                else:
                    (p * g).sum().backward()

        # ---- window 1: one no_sync micro-batch, then an exchanged one; counters travel as ONE collective
        st.zero_grad()
        g1, g2 = oracle_grads(10 + rank), oracle_grads(20 + rank)
        with st.no_sync():
            run_backward(g1)
            ok = ok and len(red.handles) == 0 and red.next_bucket == 0
        run_backward(g2)
        ok = ok and red.next_bucket == len(red.buckets)      # every bucket went out during the pass, in order
        counts = red.all_reduce_counts(torch.tensor([2 + rank, 100 * (rank + 1)], dtype=torch.int64))
        red.finish()
        ok = ok and counts.tolist() == [sum(2 + r for r in range(world)), sum(100 * (r + 1) for r in range(world))]
        want = torch.zeros_like(fs.flat_g)
        for r in range(world):
            for gr in (oracle_grads(10 + r), oracle_grads(20 + r)):
                for p, o in zip(fs.params, fs.offsets):
                    want[o:o + p.numel()] += gr[names[id(p)]].reshape(-1)
        ok = ok and torch.allclose(fs.flat_g, want, rtol=1e-5, atol=1e-7)
        ok = ok and red.n_touched == len(fs.params) and st._active_runs() == [(0, fs.total, 1)]
        # ---- window 2: frozen feature extractor + dropped encoder layer 1 -> those ranges get no update and keep their step
        fs.flat_g.zero_()
        st._grads_clean = True
        st.zero_grad()
        frozen = ("feature_extractor.", "encoder.transformer.encoders.1.")
        run_backward(g2, skip=frozen)
        red.finish()
        runs = st._active_runs()
        covered = torch.zeros(fs.total, dtype=torch.bool)
        for lo, hi, nstep in runs:
            covered[lo:hi] = True
            ok = ok and nstep == 2
        for i, (p, o) in enumerate(zip(fs.params, fs.offsets)):
            is_frozen = any(names[id(p)].startswith(s) for s in frozen)
            ok = ok and bool(covered[o:o + p.numel()].all()) == (not is_frozen)
            ok = ok and st.param_steps[i] == (1 if is_frozen else 2)
        ok = ok and len(runs) >= 2
        def chk(ok_, tag, cond):
            if not cond:
                print("rank %d: DP contract check %s failed" % (rank, tag), flush=True)
            return ok_ and bool(cond)

        # ---- window 3: layer drop drawn per rank (each process has its own numpy RNG, audio8/wav2vec2.py:613-622): rank 0 drops
        # encoder layer 0, rank 1 drops layer 1.  The bucket holding a dropped layer never completes on that rank and holds the
        # later ones back until the step: both ranks must still issue the same collective sequence (buckets in index order, then
        # the counts + flags collective), end with the SUM, and agree on which parameters to update (the union).
        fs.flat_g.zero_()
        st._grads_clean = True
        st.zero_grad()
        st.sync_touched = True
        mine = "encoder.transformer.encoders.%d." % rank
        run_backward(g2, skip=(mine,))
        ok = chk(ok, 1, red.next_bucket < len(red.buckets))  # something was held back on this rank
        local_touched = list(red.touched)
        ok = chk(ok, 2, not all(local_touched))
        try:                                                   # the counts collective may not overtake pending buckets
            red.all_reduce_counts(torch.zeros(2, dtype=torch.int64))
            ok = chk(ok, 99, False)
        except RuntimeError:
            pass
        counts = st._exchange(batch_size=3 + rank, num_tokens=torch.tensor(7 * (rank + 1)))
        ok = chk(ok, 3, counts.tolist() == [sum(3 + r for r in range(world)), sum(7 * (r + 1) for r in range(world))])
        ok = chk(ok, 4, all(red.touched) and red.n_touched == len(fs.params))  # union over ranks: every layer was used somewhere
        want = torch.zeros_like(fs.flat_g)
        for r in range(world):
            gr = oracle_grads(20 + r)
            for p, o in zip(fs.params, fs.offsets):
                if not names[id(p)].startswith("encoder.transformer.encoders.%d." % r):
                    want[o:o + p.numel()] += gr[names[id(p)]].reshape(-1)
        ok = chk(ok, 5, torch.allclose(fs.flat_g, want, rtol=1e-5, atol=1e-7))
        both = [torch.empty_like(fs.flat_g) for _ in range(world)]
        dist.all_gather(both, fs.flat_g)
        ok = chk(ok, 6, torch.equal(both[0], both[1]))  # bitwise the same reduced gradients on both ranks
        ok = chk(ok, 7, st._exchange() is None)  # idempotent: nothing is issued twice
        # ---- window 4: the unchanged drivers' order -- backward; clip_grad_norm_; step (audio8/pretrain.py:179-183): the patched
        # clip first completes the exchange and forms the MEAN, so every rank clips the same, averaged, gradient
        from audio8_amd import driver
        fs.flat_g.zero_()
        st._grads_clean = True
        st.zero_grad()
        st.sync_touched = False
        run_backward(g2)
        total = driver.clip_grad_norm_(model.parameters(), 0.5)
        mean = torch.zeros_like(fs.flat_g)
        for r in range(world):
            gr = oracle_grads(20 + r)
            for p, o in zip(fs.params, fs.offsets):
                mean[o:o + p.numel()] += gr[names[id(p)]].reshape(-1) / world
        mnorm = float(mean.double().norm())                    # (float32 norm() of 24 M elements on the CPU is only good to ~2e-4)
        ok = chk(ok, 8, st._mean_applied and abs(float(total) - mnorm) <= 1e-4 * mnorm)
        ok = chk(ok, 9, torch.allclose(fs.flat_g, mean * min(1.0, 0.5 / (mnorm + 1e-6)), rtol=1e-4, atol=1e-8))
        st.sync_grads()                                        # a second call must not divide again
        ok = chk(ok, 10, torch.allclose(fs.flat_g, mean * min(1.0, 0.5 / (mnorm + 1e-6)), rtol=1e-4, atol=1e-8))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_wav2vec2_module_dp_contract_gloo_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_w2v_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


# ---------------------------------------------------------------------------------------------------------------------------
# bench.py's launcher contract at N = 8 (the driver's scaling run), rehearsed on CPU ranks: `--dry-run` takes the same path through
# spawn_ranks / torch.distributed.run / the 127.0.0.1 rendezvous, but the ranks only join a gloo group (eight processes may not share
# the single test GPU).
def _bench(args, env_extra=None, timeout=300):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="1", **(env_extra or {}))
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout, cwd=root)


def test_bench_gpus_8_dry_run_is_eight_ranks_and_one_json_line():
    import json
    r = _bench(["--gpus", "8", "--dry-run"])
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["rank_sum"] == 36.0 and out["config"]["global_batch"] == 64 and out["config"]["parallelism"] == "dp8"


@pytest.mark.parametrize("form", ["--host-cores 2", "--host-cores=2"])
def test_bench_launcher_leaves_every_rank_its_own_host_slice(form):
    """the launcher used to pin ITSELF to four cores; its ranks inherited that mask and sliced it again, so all ranks sat on
    the same cores.  Two ranks started through spawn_ranks must end up on disjoint masks of the requested size, whichever way argparse
    would accept the flag, and the launcher's own mask must reach them whole (A8_HOST_CPUS)."""
    import json
    cpus = sorted(os.sched_getaffinity(0))
    if len(cpus) < 4:
        pytest.skip("needs >= 4 CPUs")
    r = _bench(["--gpus", "2", "--dry-run"] + form.split())
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    m0, m1 = out["config"]["rank_cpu_masks"]
    assert out["config"]["host_cores"] == 2 and len(m0) == 2 and len(m1) == 2, out
    assert not set(m0) & set(m1), out
    half = len(cpus) // 2
    assert set(m0) <= set(cpus[:half]) and set(m1) <= set(cpus[half:2 * half]), (out, cpus)


def test_host_slice_rule():
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    argv, sys.argv = sys.argv, ["bench.py", "--host-cores", "0"]           # importing bench.py runs its placement prelude: keep it off
    try:
        spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
        b = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(b)
    finally:
        sys.argv = argv
    cpus = list(range(100, 116))
    sl = [b.host_slice(cpus, 8, r, 4) for r in range(8)]
    assert all(len(x) == 2 for x in sl) and sorted(sum(sl, [])) == cpus            # 16 CPUs / 8 ranks: two each, a partition
    assert [b.host_slice(cpus, 1, 0, 4), b.host_slice(cpus, 2, 1, 4)] == [[100, 101, 102, 103], [108, 109, 110, 111]]
    assert [b.host_slice([7, 9], 4, r, 4) for r in range(4)] == [[7], [9], [7], [9]]   # more ranks than CPUs: shared round-robin, never empty


@pytest.mark.parametrize("when", ["A8_BENCH_FAIL_BEFORE_INIT", "A8_BENCH_FAIL_AFTER_INIT"])
def test_bench_rank_failure_makes_the_launcher_exit_nonzero(when):
    """One rank raising -- before it reaches init_process_group, or after the group exists -- must end the whole launch with a
    non-zero exit code instead of leaving seven ranks waiting in the rendezvous (fresh child processes only: the launcher itself
    never initialises a GPU and never re-executes itself)."""
    r = _bench(["--gpus", "8", "--dry-run"], {when: "3"}, timeout=240)
    assert r.returncode != 0
    assert "injected failure of rank 3" in (r.stderr + r.stdout)
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
