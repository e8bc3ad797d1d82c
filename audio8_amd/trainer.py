"""Step-level runtime either side of the hot path:

  * FlatState     -- all parameters / gradients / Adam moments live in four contiguous f32 buffers
                     (288 GB of HBM: nothing is sharded or offloaded); module parameters are views.
  * GradReducer   -- data-parallel gradient exchange: the flat gradient buffer is cut into buckets in
                     reverse-parameter order; when the last gradient of a bucket has been accumulated
                     (post-accumulate hook) the bucket's all-reduce is issued on a side HIP stream behind
                     an event, so RCCL traffic over xGMI overlaps the rest of backward.  Replaces DDP's
                     reducer for audio8/pretrain.py:152-159 (the model stays wrappable by DDP as well).
  * Stepper       -- zero_grad / clip_grad_norm_ / AdamW of audio8/pretrain.py:171-186 as two kernel
                     launches (global sum of squares, fused clip+AdamW), no host sync.
"""
import contextlib
import os

import torch
import torch.distributed as dist

from . import _lib as L
from . import kernels as K
from . import ops


class FlatState:
    def __init__(self, model: torch.nn.Module):
        params = self._ordered(model)
        dev = params[0].device
        self.params = params
        self.offsets, off = [], 0
        for p in params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4             # keep every view 16-byte aligned
        self.total = off
        self.flat_p = torch.zeros(off, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(off, device=dev, dtype=torch.float32)
        self.flat_m = torch.zeros(off, device=dev, dtype=torch.float32)
        self.flat_v = torch.zeros(off, device=dev, dtype=torch.float32)
        for p, o in zip(params, self.offsets):
            view = self.flat_p[o:o + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            p.grad = self.flat_g[o:o + p.numel()].view_as(p)

    @staticmethod
    def _ordered(model):
        """Parameters in module order, except that each attention block's w_Q/w_K/w_V weights (and biases) are
        made adjacent so the fused N = 3d QKV GEMM and its weight gradient address them as one [3d, d] matrix."""
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        by_name = dict(named)
        out, done = [], set()
        for n, p in named:
            if id(p) in done:
                continue
            if n.endswith("self_attn.w_Q.layer.weight"):
                stem = n[: -len("w_Q.layer.weight")]
                group = [stem + f"w_{c}.layer.{leaf}" for leaf in ("weight", "bias") for c in "QKV"]
                if all(g in by_name for g in group):
                    for g in group:
                        out.append(by_name[g])
                        done.add(id(by_name[g]))
                    continue
            out.append(p)
            done.add(id(p))
        # The flat layout is cut into exchange buckets from its END, and buckets leave strictly in that order: a parameter whose gradient
        # arrives late in backward must not sit near the end.  AudioTransformerEncoder registers its input LayerNorm (`ln`, applied BEFORE
        # the layer stack: audio8/wav2vec2.py:623,639) AFTER the stack, so its gradient -- which arrives once the whole stack's backward is
        # through -- held the first bucket, and with it every bucket, until 70 % of backward was over (round 6, tools/exchange_exposure.py:
        # all twelve buckets left in a burst at 9.0 of 11.3 ms).  Laid out in forward-use order (in front of the stack's parameters) the
        # buckets leave layer by layer from the start of backward.  (DDP gets the same effect by rebuilding its buckets in the order the
        # gradients arrived during the first iteration.)
        names = {id(p): n for n, p in named}
        stack_first = {}
        for k, p in enumerate(out):
            n = names[id(p)]
            if ".transformer.encoders." in n or n.startswith("transformer.encoders."):
                stack_first.setdefault(n.split("transformer.encoders.")[0], k)
        for prefix, first in sorted(stack_first.items(), key=lambda kv: -kv[1]):
            late = [k for k, p in enumerate(out) if names[id(p)] in (prefix + "ln.weight", prefix + "ln.bias")]
            if late and all(k > first for k in late):
                moved = [out[k] for k in late]
                rest = [p for k, p in enumerate(out) if k not in late]
                k0 = next(i for i, p in enumerate(rest) if names[id(p)].startswith(prefix + "transformer.encoders."))
                out = rest[:k0] + moved + rest[k0:]
        return out

    def zero_grad(self, already_clean=False):
        if not already_clean:
            self.flat_g.zero_()
        for p, o in zip(self.params, self.offsets):      # keep .grad bound to the flat views
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                p.grad = self.flat_g[o:o + p.numel()].view_as(p)


class WeightCache:
    """Compute-dtype working copies of every GEMM weight, rebuilt from the f32 masters by ONE kernel launch per
    optimizer step (a8_prepare_weights_device):
        ops.WEIGHT_CACHE[id(w)]      = (W2 [rows, cols], W2^T [cols, rows])      Linear / fused QKV / conv [co][kk*cin+ci]
        ops.CONV_DGRAD_CACHE[id(w)]  = (B_even [cin, 2*co], B_odd [cin, co])     k = 3, stride 2 conv data gradient
    """

    def __init__(self, model: torch.nn.Module, dtype=torch.bfloat16):
        import ctypes as C
        import numpy as np
        from .layers import MultiHeadedAttention
        self.dtype = dtype
        dev = next(model.parameters()).device
        jobs = []               # (src_param, rows, cols, kw, tap, dst, dstT, ld_dst, ld_dstT)
        early_ids = {id(p) for n, p in model.named_parameters() if n.startswith(("feature_extractor.", "proj_to_input.", "encoder.feature_extractor.", "encoder.proj_to_input."))}
        self.cache, self.conv_dgrad = {}, {}
        self._keep = []

        def alloc(*shape):
            t = torch.empty(shape, device=dev, dtype=dtype)
            self._keep.append(t)
            return t

        fused = set()
        for m in model.modules():
            if isinstance(m, MultiHeadedAttention):
                wq, wk, wv = m.w_Q.layer.weight, m.w_K.layer.weight, m.w_V.layer.weight
                if ops._adjacent(wq, wk, wv):
                    r, c = 3 * wq.shape[0], wq.shape[1]
                    dst, dst_t = alloc(r, c), alloc(c, r)
                    jobs.append((wq, r, c, 1, -1, dst, dst_t, 0, 0))
                    self.cache[id(wq)] = (dst, dst_t)
                    fused.update((id(wq), id(wk), id(wv)))
        for n, p in model.named_parameters():
            if id(p) in fused or not p.requires_grad or "pos_conv" in n:
                continue
            if n.endswith("quantizer.vars"):              # codebook [1, G*V, D]: a 2-D working copy for the straight-through product
                r, c = p.shape[1], p.shape[2]
                dst = alloc(r, c)
                jobs.append((p, r, c, 1, -1, dst, None, 0, 0))
                self.cache[id(p)] = (dst, None)
                continue
            if p.dim() == 2:
                r, c = p.shape
                dst, dst_t = alloc(r, c), alloc(c, r)
                jobs.append((p, r, c, 1, -1, dst, dst_t, 0, 0))
                self.cache[id(p)] = (dst, dst_t)
            elif p.dim() == 3 and p.shape[1] > 1:
                co, cin, kw = p.shape
                dst, dst_t = alloc(co, kw * cin), alloc(kw * cin, co)
                jobs.append((p, co, kw * cin, kw, -1, dst, dst_t, 0, 0))
                self.cache[id(p)] = (dst, dst_t)
                if kw == 3:
                    b_even, b_odd = alloc(cin, 2 * co), alloc(cin, co)
                    # dx[2t]   = dz[t-1] W[:, :, 2] + dz[t] W[:, :, 0]  ->  B_even[ci] = [W[:, ci, 2] | W[:, ci, 0]]
                    jobs.append((p, co, cin, kw, 2, None, b_even, 0, 2 * co))
                    jobs.append((p, co, cin, kw, 0, None, b_even[:, co:], 0, 2 * co))
                    # dx[2t+1] = dz[t] W[:, :, 1]
                    jobs.append((p, co, cin, kw, 1, None, b_odd, 0, 0))
                    self.conv_dgrad[id(p)] = (b_even, b_odd)
        # Two descriptor tables: what the forward pass needs first (conv feature extractor, input projection) and the rest (encoder
        # stack, heads).  refresh() builds the first on the caller's stream and -- given a side stream -- the second beside it, so the
        # 0.16 ms of the second launch run under the conv stack; the encoder stack waits for ops.WPREP_EVENT once.
        self.params = [j[0] for j in jobs]
        self._src_ptrs = [p.data_ptr() for p in self.params]
        self.tables = [self._table([j for j in jobs if id(j[0]) in early_ids], dev), self._table([j for j in jobs if id(j[0]) not in early_ids], dev)]
        self.side = None
        self.early_ids = frozenset(early_ids)

    @staticmethod
    def _table(jobs, dev):
        import ctypes as C
        import numpy as np
        n = len(jobs)
        if n == 0:
            return None
        descs = (L.WPrepDesc * n)()
        starts = [0]
        for i, (src, r, c, kw, tap, dst, dst_t, ld, ld_t) in enumerate(jobs):
            descs[i] = L.WPrepDesc(src.data_ptr(), dst.data_ptr() if dst is not None else None,
                                   dst_t.data_ptr() if dst_t is not None else None, r, c, kw, tap, ld, ld_t)
            starts.append(starts[-1] + ((r + 63) // 64) * ((c + 63) // 64))
        raw = np.frombuffer(C.string_at(C.addressof(descs), C.sizeof(descs)), dtype=np.uint8).copy()
        return (torch.from_numpy(raw).to(dev), torch.tensor(starts, dtype=torch.int32, device=dev), n, starts[-1])

    def _launch(self, table):
        descs_dev, starts_dev, n, tiles = table
        L.check(L.lib().a8_prepare_weights_device(L.dt(self._keep[0]), L.ptr(descs_dev), L.ptr(starts_dev), n, tiles, L.stream()),
                "a8_prepare_weights_device")

    def refresh(self):
        for i, p in enumerate(self.params):               # parameters may have been re-pointed (e.g. .to(), load)
            if p.data_ptr() != self._src_ptrs[i]:
                raise RuntimeError("WeightCache: a parameter's storage moved; rebuild the Stepper")
        early, late = self.tables
        if early is not None:
            self._launch(early)
        if late is None:
            return
        if self.side is None or not ops.WPREP_SPLIT:
            self._launch(late)
            return
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(cur)                                    # the masters are final (AdamW is enqueued on `cur`)
        self.side.wait_event(ev)
        with torch.cuda.stream(self.side):
            self._launch(late)
            done = torch.cuda.Event()
            done.record(self.side)
        ops.WPREP_EVENT = done                            # taken (and cleared) by ops.wait_weight_prep() in front of the first consumer


WAIT_ALL_STREAMS = os.environ.get("A8_BUCKET_WAIT_ALL", "0") == "1"       # developer A/B: every bucket waits for every stream (round 5)


class GradReducer:
    """Bucketed all-reduce of the flat gradient buffer, overlapped with backward."""

    def __init__(self, state: FlatState, bucket_mb: float = 32.0, process_group=None, tail_elems: int = 65536, payload=None):
        self.state = state
        self.pg = process_group
        # Payload of the exchange (380 MB of f32 per step is ~4.3 ms on one xGMI link against an ~11 ms step).
        # "f32" (default): the flat f32 gradient buffer is all-reduced in place.  "bf16": each bucket is cast into a bf16 staging buffer,
        # all-reduced there (half the bytes on the links) and cast back over the bucket's f32 slot -- every rank ends with the same SUM,
        # exact to bf16 rounding of the addends and of the ring's partial sums (relative 2^-8 per element; the optimizer's moments and
        # the masters stay f32).  Off by default: the reference exchanges f32 (DDP over f32 parameters, audio8/pretrain.py:158).
        payload = payload or os.environ.get("A8_GRAD_PAYLOAD", "f32")
        if payload not in ("f32", "bf16"):
            raise ValueError(f"a8hip GradReducer: payload must be 'f32' or 'bf16', got {payload!r}")
        self.payload = payload
        self.stage = None                    # bf16 staging buffer, one element per gradient element (allocated on first use)
        self.deferred = []                   # (handle, f32 view, bf16 view): cast-backs of a backend that is not stream ordered (gloo)
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # A8_FORCE_REDUCER=1: run the whole exchange path (hooks, buckets, side stream, RCCL calls) on a single rank too --
        # a rehearsal of the N > 1 code on a one-GPU box (an all-reduce over one rank leaves the data unchanged)
        self.enabled = self.world > 1 or (dist.is_initialized() and os.environ.get("A8_FORCE_REDUCER", "0") == "1")
        self.on_gpu = state.flat_g.is_cuda
        self.side = torch.cuda.Stream() if self.on_gpu else None     # exchange + per-bucket gradient-norm stream
        self._dev_index = state.flat_g.device.index if self.on_gpu else None
        self._wgrad_raw = None              # raw handle of the weight-gradient stream (set with extra_streams by the Stepper)
        # a8_set_tuning key 8 (the conv layers' persistent stream GEMMs launched as n CUs' worths of blocks: shorter static shares, the
        # surplus dispatched as CUs free up -- work-conserving when RCCL kernels hold CUs at launch) costs 0.09 ms per step alone on the
        # GPU and has never been measured under real RCCL traffic: it stays at the library default (1) unless A8_STREAM_GRID=<n> asks;
        # close() restores the value found here
        self._stream_grid_saved = None
        if self.on_gpu and os.environ.get("A8_STREAM_GRID", "") != "":
            import ctypes
            from . import _lib as L
            found = ctypes.c_int(0)
            L.check(L.lib().a8_get_tuning(8, ctypes.addressof(found)), "a8_get_tuning")
            self._stream_grid_saved = int(found.value)               # (whatever the caller or A8_TUNE had set: restored by Stepper.close())
            L.check(L.lib().a8_set_tuning(8, int(os.environ["A8_STREAM_GRID"])), "a8_set_tuning")
        # RCCL enqueues a collective on the stream it is issued from, so kernels behind it on that stream see the reduced data;
        # gloo (CPU rehearsals of the GPU path) completes on its own threads: only wait() orders it
        self.stream_ordered = True
        if self.enabled:
            try:
                self.stream_ordered = dist.get_backend(process_group) == "nccl"
            except Exception:                # (a stubbed collective in tests: runs on the issuing stream)
                pass
        self.norm_sq = None                  # f32 [1] device accumulator of sum g^2 (set by the Stepper): each bucket adds its share as it
                                             # completes (after its all-reduce), under the rest of backward
        # Exposure measurement on ONE GPU (tools/exchange_exposure.py): every bucket's collective is followed -- on the exchange stream, where
        # RCCL's kernel would run -- by a one-wave delay of bytes / fake_gbps (a8_delay_us), so the step time shows how much of an exchange
        # of that speed is NOT hidden under backward.  `trace` (a list, when set) receives (bucket, start event, end event, bytes) per bucket.
        fake = os.environ.get("A8_FAKE_EXCHANGE_GBPS", "")
        self.fake_gbps = float(fake) if fake else 0.0
        self.trace = None
        cap = int(bucket_mb * 1024 * 1024 / 4)
        # buckets over contiguous flat ranges; built from the LAST parameter backwards (gradients of late
        # layers are ready first), each bucket = [lo, hi) in elements plus the parameter indices it holds
        self.buckets = []
        hi, members = state.total, []
        for i in range(len(state.params) - 1, -1, -1):
            members.append(i)
            lo = state.offsets[i]
            if hi - lo >= cap or i == 0:
                self.buckets.append({"lo": lo, "hi": hi, "members": members, "pending": len(members)})
                hi, members = lo, []
        # The parameters whose gradients arrive LAST (the first ones: conv layer 0 and its GroupNorm, a few thousand
        # elements) get a bucket of their own: the big final bucket then goes out while conv layer 0's backward still runs,
        # and only a latency-sized exchange is left exposed after the last backward kernel.
        last = self.buckets[-1]
        idx = sorted(last["members"])
        n_tail, acc = 0, 0
        for i in idx:
            size = (state.offsets[i + 1] if i + 1 < len(state.params) else state.total) - state.offsets[i]
            if acc + size > tail_elems:
                break
            acc += size
            n_tail += 1
        if 0 < n_tail < len(idx):
            cut = state.offsets[idx[n_tail]]
            self.buckets[-1] = {"lo": cut, "hi": last["hi"], "members": idx[n_tail:], "pending": len(idx) - n_tail}
            self.buckets.append({"lo": last["lo"], "hi": cut, "members": idx[:n_tail], "pending": n_tail})
        self.bucket_of = {}
        for b, bk in enumerate(self.buckets):
            for i in bk["members"]:
                self.bucket_of[i] = b
        self.handles = []
        self.next_bucket = 0
        self.marked = [False] * len(state.params)
        # touched[i]: parameter i received a gradient since the last optimizer step (any backward pass of the accumulation
        # window).  The optimizer skips the others, like torch's AdamW skips `p.grad is None` (frozen encoder / feature
        # extractor of audio8/train.py:296-297, layer-dropped blocks).
        self.touched = [False] * len(state.params)
        self.n_touched = 0
        self.suspended = False              # no_sync(): gradients accumulate locally, nothing is exchanged
        self.extra_streams = []
        self.compute_stream = None
        self.index_of = {id(p): i for i, p in enumerate(state.params)}
        # Gradient arrival is reported twice over: backward kernels that accumulate in place call param_ready() (ops.GRAD_READY_HOOK);
        # gradients that come through autograd's AccumulateGrad fire a post-accumulate hook.  The hooks are Python callbacks on the
        # autograd thread (~1 ms per step for wav2vec2-base's 210 parameters), so after the first exchanged step they are kept only
        # on the parameters that did NOT report through param_ready() (prune_hooks()).
        self.direct_seen = [False] * len(state.params)
        self.hook_handles = [p.register_post_accumulate_grad_hook(self._make_hook(i)) for i, p in enumerate(state.params)]
        self.hooks_pruned = 0

    def _mark(self, i):
        if not self.touched[i]:
            self.touched[i] = True
            self.n_touched += 1
        if self.suspended or not self._active():
            return
        # idempotent per backward pass: a parameter whose gradient a backward kernel wrote directly is reported through
        # ops.GRAD_READY_HOOK, and autograd's post-accumulate hook fires for it as well (with an undefined gradient);
        # counting both would release the bucket before its other members' gradients are enqueued
        if self.marked[i]:
            return
        self.marked[i] = True
        bk = self.buckets[self.bucket_of[i]]
        bk["pending"] -= 1
        if self.on_gpu:
            # the stream(s) this gradient was enqueued on: the caller's current stream (backward kernels report from the context they launched
            # in -- the weight-gradient stream for almost every parameter) and, while a split backward is in flight outside that stream,
            # the half-batch stream too.  The bucket's collective waits for exactly these (see _launch).
            raw = torch._C._cuda_getCurrentRawStream(self._dev_index)      # (an int: a torch.cuda.current_stream() object per parameter is 1 ms per step)
            bs = bk.setdefault("streams", {})
            if raw not in bs:
                bs[raw] = torch.cuda.current_stream()
            aux = ops._BWD_AUX_LIVE
            if aux is not None and raw != self._wgrad_raw:
                bs[aux.cuda_stream] = aux
        # Collectives must be issued in the same order on every rank: buckets go out strictly in index order (the order in
        # which backward completes them anyway); one whose gradients arrive early waits for its predecessors, one that
        # never completes (frozen / dropped layers) holds the rest back until finish().
        while self.next_bucket < len(self.buckets) and self.buckets[self.next_bucket]["pending"] == 0:
            self._launch(self.buckets[self.next_bucket])
            self.next_bucket += 1

    def _active(self):
        """The bucket schedule runs: there is an exchange, a per-bucket gradient norm, or a stand-in for the exchange's transport time."""
        return self.enabled or self.norm_sq is not None or self.fake_gbps > 0.0 or self.trace is not None

    def _make_hook(self, i):
        def hook(_p):
            self._mark(i)
        return hook

    def first_touch(self, p):
        """ops.GRAD_FIRST_TOUCH: p's slot of the flat gradient buffer still holds the zeros of the last zero_grad()."""
        i = self.index_of.get(id(p))
        return i is not None and not self.touched[i]

    def param_ready(self, p):
        """ops.GRAD_READY_HOOK: a backward kernel has enqueued p's gradient directly into the flat buffer."""
        i = self.index_of.get(id(p))
        if i is not None:
            self.direct_seen[i] = True
            self._mark(i)

    def prune_hooks(self):
        """Drop the autograd hooks of parameters whose gradients are announced by the backward kernels themselves."""
        n = sum(self.direct_seen)
        if n == self.hooks_pruned:               # nothing new since the last call (parameters unfrozen later report then)
            return
        for i, h in enumerate(self.hook_handles):
            if h is not None and self.direct_seen[i]:
                h.remove()
                self.hook_handles[i] = None
        self.hooks_pruned = n

    def rearm_hooks(self):
        """Re-register every hook (call after changing which gradients are accumulated in place, e.g. another compute dtype)."""
        for i, p in enumerate(self.state.params):
            if self.hook_handles[i] is None:
                self.hook_handles[i] = p.register_post_accumulate_grad_hook(self._make_hook(i))
        self.direct_seen = [False] * len(self.state.params)
        self.hooks_pruned = 0

    def _launch(self, bk):
        view = self.state.flat_g[bk["lo"]:bk["hi"]]
        h = None
        if self.on_gpu:
            # The exchange stream waits for the streams this bucket's gradients were enqueued on -- recorded per parameter in _mark() -- and
            # for nothing else (round 5 recorded an event on the compute, half-batch and weight-gradient streams for every bucket "to be
            # safe"; A8_BUCKET_WAIT_ALL=1 restores that set).  What the schedule costs on one GPU, 0.35-0.45 ms per step for twelve buckets,
            # is almost all the event RECORD on the weight-gradient stream (tools/bucket_probe.py: bookkeeping 0, stand-in kernels 0, records
            # alone 0.30 ms, waits + final join 0.06; device-scope events, another place in the layer's launch order: no change).
            streams = dict(bk.get("streams", {}))
            if WAIT_ALL_STREAMS:
                cur = torch.cuda.current_stream()
                streams[cur.cuda_stream] = cur
                for s2 in [self.compute_stream] + list(self.extra_streams):
                    if s2 is not None:
                        streams[s2.cuda_stream] = s2
            for st in streams.values():
                ev = torch.cuda.Event()
                ev.record(st)
                self.side.wait_event(ev)
            with torch.cuda.stream(self.side):
                if self.trace is not None:
                    ev0 = torch.cuda.Event(enable_timing=True)
                    ev0.record(self.side)
                if self.fake_gbps > 0.0:
                    nbytes = (bk["hi"] - bk["lo"]) * (2 if self.payload == "bf16" else 4)
                    from . import _lib as L
                    L.check(L.lib().a8_delay_us(nbytes / (self.fake_gbps * 1e3), self.side.cuda_stream), "a8_delay_us")
                # RCCL: the BLOCKING form (async_op=False).  Since torch 2.7 ProcessGroupNCCL enqueues a blocking-form collective on the CURRENT
                # stream -- here the exchange stream -- and returns without waiting on the host; the async form runs on a stream of its own and
                # hands back a handle.  That fifth stream matters on this hardware: HIP multiplexes streams over four hardware queues, the step
                # already uses four (caller, half-batch, weight-gradient, exchange), so the collective's own stream would share a queue with one
                # of them and its kernels would run IN ORDER with that queue's compute launches (profiles/r06_stream_priority.txt: a fifth
                # hardware queue is worse still).  On the exchange stream the collectives sit where tools/exchange_exposure.py measured them.
                # (An older torch runs the blocking form on its own stream and makes the current stream wait for it: same results, same order.)
                if self.enabled and self.payload == "bf16":
                    sv = self._stage_view(bk)
                    sv.copy_(view)                                            # f32 -> bf16 (round to nearest even)
                    if self.stream_ordered:
                        dist.all_reduce(sv, op=dist.ReduceOp.SUM, group=self.pg, async_op=False)
                        view.copy_(sv)                                        # bf16 SUM over ranks back into the f32 slot
                    else:
                        self.deferred.append((dist.all_reduce(sv, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), view, sv))
                elif self.enabled and self.stream_ordered:
                    dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=False)
                elif self.enabled:
                    h = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                if self.norm_sq is not None and self.stream_ordered:      # stream order: behind this bucket's collective
                    K.sumsq_(view, self.norm_sq)
                if self.trace is not None:
                    ev1 = torch.cuda.Event(enable_timing=True)
                    ev1.record(self.side)
                    self.trace.append((self.buckets.index(bk), ev0, ev1, (bk["hi"] - bk["lo"]) * (2 if self.payload == "bf16" else 4)))
        elif self.enabled and self.payload == "bf16":    # gloo on CPU (tests)
            sv = self._stage_view(bk)
            sv.copy_(view)
            self.deferred.append((dist.all_reduce(sv, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), view, sv))
        elif self.enabled:                               # gloo on CPU (tests): same bucket schedule, no streams
            h = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        if h is not None:
            self.handles.append(h)
        bk["launched"] = True

    def _stage_view(self, bk):
        if self.stage is None:
            self.stage = torch.empty(self.state.total, device=self.state.flat_g.device, dtype=torch.bfloat16)
        return self.stage[bk["lo"]:bk["hi"]]

    def begin_pass(self):
        """Before a backward pass whose gradients will be exchanged: every bucket waits for all of its members again."""
        if self.on_gpu:
            self.compute_stream = torch.cuda.current_stream()
        for bk in self.buckets:
            bk["pending"] = len(bk["members"])
            bk["launched"] = False
            bk["streams"] = {}
        self.marked = [False] * len(self.state.params)
        self.handles = []
        self.deferred = []
        self.next_bucket = 0

    def begin_step(self):
        """After an optimizer step (zero_grad): a new accumulation window."""
        self.touched = [False] * len(self.state.params)
        self.n_touched = 0
        self.begin_pass()

    def all_reduce_counts(self, counts):
        """SUM over ranks of a small device tensor (the batch-size / token counters of audio8/train.py:320-321 and, when the
        touched sets may differ between ranks, the per-parameter "has a gradient" flags, as ONE collective) on the exchange
        stream.  Always issued AFTER every gradient bucket of the step (launch_pending()), so all ranks issue the same sequence of
        collectives whatever each of them still had pending; wait() makes the compute stream wait for it."""
        if not self.enabled:
            return counts
        if self.next_bucket < len(self.buckets):
            raise RuntimeError("a8hip GradReducer: the counts collective must follow launch_pending() (same collective order on every rank)")
        if self.on_gpu:
            self.side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                if self.stream_ordered:              # RCCL: on the exchange stream itself (see _launch)
                    dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=self.pg, async_op=False)
                else:
                    self.handles.append(dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
            counts.record_stream(self.side)
        else:
            self.handles.append(dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        return counts

    def launch_pending(self):
        """Issue, in index order, every bucket that is not out yet: the ones whose parameters produced no gradient on this rank
        (frozen encoder, audio8/train.py:266-268; a layer this rank dropped) and everything they held back."""
        if not self._active():
            return
        for bk in self.buckets[self.next_bucket:]:       # in index order, like the ones already out
            if not bk.get("launched"):
                self._launch(bk)
        self.next_bucket = len(self.buckets)

    def wait(self):
        """Make the compute stream (and, for gloo, the host) wait for every collective issued so far.  Gradients hold the SUM
        over ranks afterwards (and norm_sq the sum of their squares)."""
        if not self._active():
            return
        for h in self.handles:
            h.wait()
        self.handles = []
        for h, view, sv in self.deferred:                 # bf16 payload over a backend that completes on its own threads
            if self.on_gpu:
                with torch.cuda.stream(self.side):        # wait() orders the CURRENT stream behind the backend's copy-back of the result:
                    h.wait()                              # it must be the stream that casts it into the f32 slot
                    view.copy_(sv)
            else:
                h.wait()
                view.copy_(sv)
        self.deferred = []
        if self.on_gpu:
            torch.cuda.current_stream().wait_stream(self.side)
            if self.norm_sq is not None and not self.stream_ordered:
                K.sumsq_(self.state.flat_g, self.norm_sq)         # (gloo rehearsals: the reduced values exist only after wait())

    def finish(self):
        """launch_pending() + wait()."""
        self.launch_pending()
        self.wait()


ACTIVE_STEPPER = None        # the Stepper that currently owns audio8_amd.ops' global hooks (driver.PassThroughDDP.no_sync finds it here)


class Stepper:
    """zero_grad / clip_grad_norm_(max_norm) / AdamW on the flat buffers, plus the data-parallel contract of the two drivers:

      pretrain.py (audio8/pretrain.py:171-186)   zero_grad(); backward; step()            gradients = mean over ranks (DDP)
      train.py    (audio8/train.py:292-326)      with no_sync(): backward   (grad_accum - 1 times)
                                                 backward; step(batch_size=, num_tokens=)
                                                 gradients = SUM over ranks and micro-batches / all-reduced batch size
                                                 (DDP's mean followed by `scale_grads(num_gpus / batch_size)`)

    Parameters that received no gradient since the last step (frozen encoder / feature extractor, layer-dropped blocks) are
    left untouched and keep their own step count, as torch's AdamW does for `p.grad is None`."""

    def __init__(self, model, lr=2.0e-4, weight_decay=1.0e-2, betas=(0.9, 0.999), eps=1e-8, clip=1.0, bucket_mb=32.0, lr_function=None,
                 wgrad_side_stream=True, process_group=None, global_step=0, sync_touched=None, grad_payload=None):
        self.state = FlatState(model)
        self.reducer = GradReducer(self.state, bucket_mb, process_group, payload=grad_payload)     # grad_payload: "f32" (default) / "bf16"
        if self.reducer.enabled:                         # C1 of SURVEY 2.2: rank 0's parameters everywhere -- BEFORE anything
            dist.broadcast(self.state.flat_p, src=0, group=process_group)      # is derived from them (the bf16 working copies below)
        self.lr, self.wd, self.betas, self.eps, self.clip = lr, weight_decay, betas, eps, clip
        self.lr_function = lr_function
        self.global_step = global_step
        self.param_steps = [0] * len(self.state.params)
        self._uniform_steps = True
        self.gnorm_sq = torch.zeros(1, device=self.state.flat_p.device, dtype=torch.float32)
        if self.state.flat_p.is_cuda and os.environ.get("A8_BUCKET_NORM", "0") != "0":
            self.reducer.norm_sq = self.gnorm_sq         # sum g^2 accumulated bucket by bucket under backward (no 380 MB pass at the end)
        self._grads_clean = False
        self._grad_ptrs = [self.state.flat_g.data_ptr() + 4 * o for o in self.state.offsets]
        ops.DIRECT_GRADS = True
        ops.GRAD_READY_HOOK = self.reducer.param_ready
        ops.GRAD_FIRST_TOUCH = self.reducer.first_touch
        self.wgrad_stream = torch.cuda.Stream() if (wgrad_side_stream and self.state.flat_p.is_cuda) else None
        ops.WGRAD_SIDE_STREAM = self.wgrad_stream
        # (the half-batch stream too: a split layer backward leaves the second halves' bias / LayerNorm gradients in flight there)
        self.reducer.extra_streams = [self.wgrad_stream, torch.cuda.current_stream() if self.state.flat_p.is_cuda else None,
                                      ops.half_stream(self.state.flat_p.device) if self.state.flat_p.is_cuda else None]
        self.reducer._wgrad_raw = self.wgrad_stream.cuda_stream if self.wgrad_stream is not None else None
        self.wcache = None
        if self.state.flat_p.is_cuda and ops.compute_dtype() == torch.bfloat16:
            self.wcache = WeightCache(model, torch.bfloat16)
            self.wcache.side = self.wgrad_stream
            self.wcache.refresh()
            ops.wait_weight_prep()                        # construction returns with the caller's stream ordered behind every copy
            ops.WEIGHT_CACHE, ops.WEIGHT_CACHE_DTYPE = self.wcache.cache, torch.bfloat16
            ops.CONV_DGRAD_CACHE = self.wcache.conv_dgrad
            ops.WPREP_EARLY = self.wcache.early_ids
        self._versions = [p._version for p in self.state.params]
        # Which parameters got a gradient can differ BETWEEN RANKS only through layer drop (each process draws from its own
        # numpy RNG, audio8/wav2vec2.py:613-622); frozen sets follow the global step and are the same everywhere.  With it the
        # per-parameter flags are OR-ed over the ranks in the step's counts collective, so that every rank updates the same set (as
        # under DDP, where a parameter any rank used receives the averaged gradient everywhere).  None = detect from the model.
        if sync_touched is None:
            sync_touched = any(float(getattr(m, "layer_drop", 0.0) or 0.0) > 0.0 for m in model.modules())
        self.sync_touched = bool(sync_touched)
        self._exchanged = False          # this step's gradient exchange has been completed (sync_grads() / step())
        self._mean_applied = False       # ... and flat_g was turned from the SUM into the MEAN over ranks (sync_grads())
        global ACTIVE_STEPPER
        ACTIVE_STEPPER = self

    @property
    def current_lr(self):
        return self.lr_function(self.global_step) if self.lr_function else self.lr

    def refresh_weights(self):
        """Rebuild the bf16 working copies (call after changing parameters outside step(), e.g. load_state_dict)."""
        if self.wcache is not None:
            self.wcache.refresh()
            ops.wait_weight_prep()
        self._versions = [p._version for p in self.state.params]

    def zero_grad(self):
        if self.wcache is not None and any(p._version != v for p, v in zip(self.state.params, self._versions)):
            self.refresh_weights()                        # someone wrote to a parameter through torch since the last step
        self.state.zero_grad(already_clean=self._grads_clean)     # step() cleared the buffer behind its read
        self._grads_clean = False
        self._exchanged = False
        self._mean_applied = False
        if self.reducer.norm_sq is not None:
            self.gnorm_sq.zero_()
        self.reducer.begin_step()

    @contextlib.contextmanager
    def no_sync(self):
        """Backward passes inside accumulate into the flat gradient buffer without any exchange (DDP.no_sync of
        audio8/train.py:300-301); the next backward outside it exchanges the accumulated sums bucket by bucket."""
        self.reducer.suspended = True
        try:
            yield
        finally:
            self.reducer.suspended = False
            self.reducer.begin_pass()

    def close(self):
        """Detach the global hooks this Stepper installed in audio8_amd.ops."""
        ops.DIRECT_GRADS = False
        ops.GRAD_READY_HOOK = None
        ops.GRAD_FIRST_TOUCH = None
        ops.WGRAD_SIDE_STREAM = None
        ops.WEIGHT_CACHE = None
        ops.WEIGHT_CACHE_DTYPE = None
        ops.CONV_DGRAD_CACHE = None
        ops.wait_weight_prep()
        ops.WPREP_EARLY = frozenset()
        red = getattr(self, "reducer", None)
        if red is not None and getattr(red, "_stream_grid_saved", None) is not None:      # A8_STREAM_GRID override: back to the default
            from . import _lib as L
            L.check(L.lib().a8_set_tuning(8, red._stream_grid_saved), "a8_set_tuning")
            red._stream_grid_saved = None
        global ACTIVE_STEPPER
        if ACTIVE_STEPPER is self:
            ACTIVE_STEPPER = None

    def _active_runs(self):
        """Contiguous flat ranges [lo, hi) of parameters that got a gradient, each with the AdamW step count it is at."""
        st, red = self.state, self.reducer
        n = len(st.params)
        if red.n_touched == n and self._uniform_steps:
            s = self.param_steps[0] + 1
            self.param_steps = [s] * n
            return [(0, st.total, s)]
        runs, i = [], 0
        ends = st.offsets[1:] + [st.total]
        while i < n:
            if not red.touched[i]:
                i += 1
                continue
            s, j = self.param_steps[i] + 1, i
            while j < n and red.touched[j] and self.param_steps[j] + 1 == s:
                self.param_steps[j] = s
                j += 1
            runs.append((st.offsets[i], ends[j - 1], s))
            i = j
        self._uniform_steps = len(set(self.param_steps)) == 1
        return runs

    def _exchange(self, batch_size=None, num_tokens=None):
        """Complete the step's gradient exchange; idempotent until zero_grad().  Order, identical on every rank: join the
        weight-gradient stream; issue every bucket still pending, in index order; then ONE small SUM collective that carries the
        batch-size / token counters (audio8/train.py:320-321) and -- sync_touched -- the per-parameter "got a gradient" flags; wait.
        Returns the all-reduced int64 device tensor [batch_size, num_tokens] when a batch size was given, else None."""
        st, red = self.state, self.reducer
        need_flags = self.sync_touched and red.enabled and not self._exchanged
        if self._exchanged and batch_size is None:
            return None
        if not self._exchanged:
            if self.wgrad_stream is not None:            # join the weight-gradient stream before gradients are read
                torch.cuda.current_stream().wait_stream(self.wgrad_stream)
            red.launch_pending()
        payload = None
        if batch_size is not None or need_flags:
            vals = [0 if batch_size is None else batch_size, 0 if num_tokens is None else num_tokens]
            host = [0 if torch.is_tensor(v) else int(v) for v in vals]
            if need_flags:
                host += [1 if t else 0 for t in red.touched]
            if st.flat_p.is_cuda:
                payload = ops.host_to_device(host, "int64")
            else:
                payload = torch.tensor(host, dtype=torch.int64)
            for k, v in enumerate(vals):
                if torch.is_tensor(v):
                    payload[k] = v.to(device=payload.device, dtype=torch.int64).reshape(())
            payload = red.all_reduce_counts(payload)
        red.wait()
        if need_flags:
            flags = payload[2:].cpu()                    # (a host sync: only on models with layer drop under data parallelism)
            red.touched = [bool(f) for f in flags.tolist()]
            red.n_touched = sum(red.touched)
        self._exchanged = True
        return payload[:2] if batch_size is not None else None

    def sync_grads(self):
        """What DistributedDataParallel guarantees when backward() returns, on demand: every gradient final (weight-gradient stream
        joined, all buckets exchanged) and `.grad` = the MEAN over ranks.  For code that reads or rescales gradients between
        backward() and step() -- the drivers' clip_grad_norm_ and OptimizerManager.scale_grads (audio8/pretrain.py:182,
        train.py:323-324).  The native path (Stepper(clip=...).step()) never needs it: the 1 / world factor rides in AdamW."""
        if self._mean_applied:
            return
        self._exchange()
        if self.reducer.enabled and self.reducer.world > 1:
            self.state.flat_g.mul_(1.0 / self.reducer.world)
        self._mean_applied = True

    def step(self, batch_size=None, num_tokens=None):
        """Exchange what is still pending, clip, AdamW.  With `batch_size` (int or 0-dim integer tensor; `num_tokens` likewise)
        the gradients are divided by the batch size summed over ranks instead of by the world size; returns the all-reduced
        int64 device tensor [batch_size, num_tokens] (train.py logs both), else None.  The step's learning rate is
        lr_function(global_step) with global_step as it stands BEFORE this step increments it."""
        st = self.state
        for p, want in zip(st.params, self._grad_ptrs):
            g = p.grad
            if g is None or g.data_ptr() != want:
                raise RuntimeError("a8hip Stepper: a parameter's .grad no longer aliases the flat gradient buffer -- use "
                                   "stepper.zero_grad(), not model.zero_grad(set_to_none=True), between steps")
        counts = self._exchange(batch_size, num_tokens)
        lr = self.current_lr
        self.global_step += 1
        world = self.reducer.world if self.reducer.enabled else 1
        if counts is not None:                           # SUM over ranks / summed batch size
            scale = float(world) if self._mean_applied else 1.0
        else:                                            # MEAN over ranks
            scale = 1.0 if self._mean_applied else 1.0 / world
        if self.reducer.norm_sq is None or self._mean_applied:
            K.zero_(self.gnorm_sq)
            K.sumsq_(st.flat_g, self.gnorm_sq)
        for lo, hi, nstep in self._active_runs():
            K.adamw_(st.flat_p[lo:hi], st.flat_g[lo:hi], st.flat_m[lo:hi], st.flat_v[lo:hi], lr, self.betas[0], self.betas[1],
                     self.eps, self.wd, nstep, self.gnorm_sq, self.clip, scale, zero_grad=True, grad_denom=counts)
        self._grads_clean = True
        if self.wcache is not None:
            self.wcache.refresh()
        self.reducer.prune_hooks()
        return counts


def pin_host_cores(n=4, rank=None, world=None):
    """Confine this process's launching thread (and every thread it starts from now on) to `n` cores of its own slice of the host and
    give torch as many intra-op threads.  One process per GPU issues ~330 launches per 11 ms step from ONE Python thread; left unpinned
    with an OpenMP thread per visible core, that thread is measurably slower (host 9.5 vs 7.0 ms per step, step 11.9 vs 11.75 ms on a
    16-core share; bench.py does the same before importing torch).  Call before the first GPU work; returns the cores taken."""
    world = max(1, int(os.environ.get("WORLD_SIZE", "1")) if world is None else world)
    rank = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0"))) if rank is None else rank
    cpus = sorted(os.sched_getaffinity(0))
    # NB: call this in the RANK process on the mask the launcher left it -- a launcher that pinned itself first hands every rank the same
    # few cores (bench.py's launcher stays unpinned for that reason)
    per = len(cpus) // world
    mine = cpus[(rank % world) * per:(rank % world + 1) * per][:n] if per else [cpus[rank % len(cpus)]]
    os.sched_setaffinity(0, mine)
    torch.set_num_threads(len(mine))
    return mine


def init_distributed_from_env():
    """One process per GPU; rendezvous from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local % max(1, torch.cuda.device_count()))      # (more ranks than GPUs only in rehearsals)
    if (world > 1 or os.environ.get("A8_FORCE_REDUCER", "0") == "1") and not dist.is_initialized():
        backend = os.environ.get("A8_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")   # nccl = RCCL on ROCm
        dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world)
    return rank, local, world
