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
