"""Autograd surface of the hot path: coarse torch.autograd.Functions whose forward/backward are
fixed sequences of liba8hip kernel launches (audio8_amd/kernels.py).  PyTorch supplies device
memory, the stream and the autograd tape; no arithmetic of the path runs in ATen, except
O(#params)-sized weight plumbing (conv-weight re-layout views, bias concatenation outside the
flat-state layout) and the scalar composition of the loss.
"""
import math
import time

import numpy as np
import torch

from . import _lib as L
from . import kernels as K

# ------------------------------------------------------------------ global compute configuration
_COMPUTE_DTYPE = torch.bfloat16
_SEED_STATE = {"base": None, "ctr": 0}


# Layer-drop draws taken ahead of time by wav2vec2.prefetch_host_draws (None: TransformerEncoderStack draws them itself).
PLANNED_LAYER_DRAWS = None


def take_layer_draws(n):
    """The n per-layer np.random.random() values of this forward: prefetched ones if a plan is active, else fresh draws."""
    global PLANNED_LAYER_DRAWS
    d, PLANNED_LAYER_DRAWS = PLANNED_LAYER_DRAWS, None
    if d is not None:
        if len(d) != n:
            raise RuntimeError("a8hip: prefetched layer-drop draws (%d) do not match the encoder depth (%d)" % (len(d), n))
        return d
    return [np.random.random() for _ in range(n)]


def host_to_device(a, dtype):
    """numpy array -> device tensor through a pinned staging buffer (cached by torch's host allocator): the copy is
    enqueued on the current stream and never blocks the host (a pageable .to(device) does)."""
    a = np.ascontiguousarray(a, dtype=dtype)
    pin = torch.empty(a.shape, dtype=torch.from_numpy(a[:0]).dtype, pin_memory=True)
    pin.numpy()[...] = a
    return pin.to("cuda", non_blocking=True)



# bench.py's per-component roofline: when a dict, the coarse Functions bracket their launch sequences with HIP events on
# the current stream, PHASE_EVENTS[name] = [(start, end), ...]  (conv_fwd, conv0_fwd, conv_bwd, conv0_bwd, enc_fwd, enc_bwd, ...)
PHASE_EVENTS = None


class _Phase:
    __slots__ = ("name", "ev")

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.ev = torch.cuda.Event(enable_timing=True)
        self.ev.record()

    def __exit__(self, *exc):
        end = torch.cuda.Event(enable_timing=True)
        end.record()
        PHASE_EVENTS.setdefault(self.name, []).append((self.ev, end))
        return False


class _NoPhase:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_PHASE = _NoPhase()


HOST_DELAY_US = 0


def phase(name):
    return _NO_PHASE if PHASE_EVENTS is None else _Phase(name)


def set_compute_dtype(dtype):
    """torch.bfloat16 (throughput mode: bf16 activations, f32 accumulate/statistics) or
    torch.float32 (parity mode: exact-f32 MFMA everywhere)."""
    global _COMPUTE_DTYPE
    if dtype not in (torch.bfloat16, torch.float32):
        raise ValueError("compute dtype must be torch.bfloat16 or torch.float32")
    _COMPUTE_DTYPE = dtype


def compute_dtype():
    return _COMPUTE_DTYPE


def next_seed():
    """Dropout seeds: a counter stream derived from torch's seed.  Never touches numpy's global RNG,
    whose draw order belongs to mask/negative sampling."""
    st = _SEED_STATE
    if st["base"] is None:
        st["base"] = torch.initial_seed() & 0xFFFFFFFF
    st["ctr"] += 1
    return (st["base"] + st["ctr"] * 0x9E3779B1) & 0xFFFFFFFF


def reset_seed_stream():
    _SEED_STATE["base"] = None
    _SEED_STATE["ctr"] = 0


# Prepared working copies of the weights, refreshed once per optimizer step by trainer.Stepper with ONE C call:
# id(param) -> (W2 [rows, cols] compute dtype, W2^T [cols, rows]); the fused QKV matrix is keyed by the w_Q weight.
WEIGHT_CACHE = None
WEIGHT_CACHE_DTYPE = None
CONV_DGRAD_CACHE = None      # id(conv weight, k = 3) -> (B_even [cin, 2co], B_odd [cin, co]) for the fused stride-2 data gradient


def cached_weights(p, dtype):
    if WEIGHT_CACHE is not None and dtype == WEIGHT_CACHE_DTYPE:
        if WPREP_EVENT is not None and id(p) not in WPREP_EARLY:
            wait_weight_prep()                 # a copy built on the side stream: whoever asks first makes the current stream wait
        return WEIGHT_CACHE.get(id(p))
    return None


def _dgrad(dy, wc, wcT, M, n_in, k_out, **kw):
    """dx[M, n_in] = dy[M, k_out] . W[k_out, n_in]: K-contiguous NT GEMM on the prepared transpose when there is one."""
    if wcT is not None:
        return K.gemm(L.NT, dy, wcT, M, n_in, k_out, **kw)
    return K.gemm(L.NN, dy, wc, M, n_in, k_out, **kw)


def _wcast(w, dtype):
    """Working copy of an f32 master weight in the compute dtype."""
    return w.detach() if dtype == torch.float32 else K.cast(w.detach().contiguous(), dtype)


def _splitk(out_tiles, kred):
    """Split the reduction so the grid fills, but does not exceed, one round of the chip's 512 co-resident
    128x128-tile blocks (576 blocks would run as two rounds at 56 % utilisation), while every split keeps at
    least ~800 reduction rows (shorter splits are dominated by prologue + atomic epilogue; measured with
    tools/splitk_sweep.py)."""
    return max(1, min(16, SPLITK_SLOTS // max(1, out_tiles), kred // SPLITK_MIN_ROWS))


SPLITK_SLOTS, SPLITK_MIN_ROWS = 512, 800     # developer knobs for the split-K rule above


def _tiles(m, n):
    return ((m + 127) // 128) * ((n + 127) // 128)


def wgrad(dy, x, n_out, n_in, m, **kw):
    """dW[n_out, n_in] (f32) = dy[m, n_out]^T . x[m, n_in]   (split-K over m); kw: leading dims / row remaps of the operands."""
    out = torch.zeros((n_out, n_in), device=dy.device, dtype=torch.float32)
    K.gemm(L.TN, dy, x, n_out, n_in, m, out=out, splitk=_splitk(_tiles(n_out, n_in), m), **kw)
    return out


def colsum(x2d, out=None):
    """f32 column sums of a [rows, cols] activation (bias gradients); accumulates into `out` when given."""
    if out is None:
        out = torch.zeros((x2d.shape[-1],), device=x2d.device, dtype=torch.float32)
    rows = x2d.numel() // x2d.shape[-1]
    L.check(L.lib().a8_gelu_bwd_bias(L.dt(x2d), L.ptr(x2d), None, None, L.ptr(out), rows, x2d.shape[-1], L.stream()), "a8_gelu_bwd_bias")
    return out


# ------------------------------------------------------------------ split weight preparation
# trainer.WeightCache.refresh() may build the encoder / head working copies on a side stream (beside the conv stack's forward);
# the first consumer of those copies calls wait_weight_prep() once.  WPREP_SPLIT is the A/B knob.
WPREP_SPLIT = True
WPREP_EVENT = None
WPREP_EARLY = frozenset()          # ids of the parameters whose copies are built on the caller's stream


def wait_weight_prep():
    global WPREP_EVENT
    ev = WPREP_EVENT
    if ev is not None:
        torch.cuda.current_stream().wait_event(ev)
        WPREP_EVENT = None


# ------------------------------------------------------------------ direct gradient accumulation
# When enabled (audio8_amd.trainer.Stepper does), a backward kernel adds its weight gradient straight into
# the parameter's existing f32 .grad buffer (a view of the flat gradient buffer) and hands autograd `None`:
# no temporary, no AccumulateGrad "+=" pass.  GRAD_READY_HOOK(param) then tells the DP reducer that the
# parameter's gradient has been enqueued on the stream.
DIRECT_GRADS = False
GRAD_READY_HOOK = None
# Weight-gradient GEMMs do not feed the backward critical path.  With direct accumulation on, they are issued on
# a side HIP stream (behind an event on their inputs) so their small grids co-run with the data-gradient chain.
# The owner of DIRECT_GRADS (trainer.Stepper) joins the side stream before it reads the gradients.
WGRAD_SIDE_STREAM = None


def _on_side(fn, *inputs):
    """Run fn() on the wgrad side stream after everything enqueued so far on the current stream (and on the half-batch stream
    while a split backward is in flight: the operands' second halves were written there)."""
    side = WGRAD_SIDE_STREAM
    if side is None:
        if _BWD_AUX_LIVE is not None:
            torch.cuda.current_stream().wait_stream(_BWD_AUX_LIVE)
        fn()
        return
    cur = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(cur)
    side.wait_event(ev)
    if _BWD_AUX_LIVE is not None:
        side.wait_stream(_BWD_AUX_LIVE)
    with torch.cuda.stream(side):
        fn()
    for t in inputs:
        if t is not None:
            t.record_stream(side)


def _target(p):
    if DIRECT_GRADS and p is not None and p.grad is not None and p.grad.dtype == torch.float32 and p.grad.is_contiguous():
        return p.grad
    return None


# The first gradient written into a parameter's slot of a clean flat buffer may be STORED instead of atomically added (f32 atomics
# run at ~1.3 TB/s chip-wide: 64 KB per 128 x 128 tile is ~13 us of a weight-gradient block's life; a plain row store ~2 us).
# GRAD_FIRST_TOUCH(p) -> True when p has received nothing since the last zero_grad (set by trainer.Stepper; None: always accumulate).
GRAD_FIRST_TOUCH = None
WGRAD_STORE_FIRST = True      # A/B knob


def _first_touch(*ps):
    return WGRAD_STORE_FIRST and GRAD_FIRST_TOUCH is not None and all(GRAD_FIRST_TOUCH(p) for p in ps)


def _ready(*ps):
    if GRAD_READY_HOOK is not None:
        for p in ps:
            if p is not None:
                GRAD_READY_HOOK(p)


def acc_wgrad(p, dy, x, n_out, n_in, m, **kw):
    """Weight gradient of parameter p: in place into p.grad (returns None) or as a fresh tensor."""
    tgt = _target(p)
    if tgt is None:
        return wgrad(dy, x, n_out, n_in, m, **kw)

    def run():
        K.gemm(L.TN, dy, x, n_out, n_in, m, out=tgt, accumulate=True, splitk=_splitk(_tiles(n_out, n_in), m), **kw)
        _ready(p)
    _on_side(run, dy, x)
    return None


# The four weight-gradient products of an encoder layer as ONE grouped launch at the end of the layer's backward:
# 432 output tiles fill the chip without split-K (each alone needs a 3-12 way split + f32 atomics to do so).
GROUP_WGRAD = True
# q/k/v bias column sums and the positional conv's weight-gradient chain also go to the side stream (A/B knob)
SIDE_EXTRAS = True


class WgradGroup:
    """Collects (params, target, dy, x, n_out, n_in, m) and issues them together on the side stream, followed by any other
    parameter-gradient work of the layer that feeds nothing downstream (`extra` callables, e.g. the q/k/v bias column sums)."""

    def __init__(self):
        self.items = []
        self.extra = []
        self.extra_inputs = []       # tensors the extra callables read (kept alive for the side stream)

    def add(self, params, tgt, dy, x, n_out, n_in, m):
        self.items.append((params, tgt, dy, x, n_out, n_in, m))

    def flush(self):
        if not self.items and not self.extra:
            return
        items, self.items = self.items, []
        extra, self.extra = self.extra, []
        extra_inputs, self.extra_inputs = self.extra_inputs, []

        # (decided NOW, on the autograd thread, before the gradients are reported ready: no split-K in the grouped launch, so a
        # first-touch product may overwrite its zero-filled target)
        acc = [not _first_touch(*params) for params, *_ in items]

        def run():
            if items:
                K.gemm_grouped(L.TN, [(dy, x, n_out, n_in, m, dict(out=tgt, accumulate=a)) for (_, tgt, dy, x, n_out, n_in, m), a in zip(items, acc)])
                for params, *_ in items:
                    _ready(*params)
            for fn in extra:
                fn()
        _on_side(run, *[t for it in items for t in (it[2], it[3])], *extra_inputs)


def acc_wgrad_grouped(group, p, dy, x, n_out, n_in, m):
    """acc_wgrad, deferred into `group` when the gradient goes straight into p.grad."""
    tgt = _target(p)
    if group is None or tgt is None:
        return acc_wgrad(p, dy, x, n_out, n_in, m)
    group.add((p,), tgt, dy, x, n_out, n_in, m)
    return None


def acc_colsum(p, x2d):
    tgt = _target(p)
    if tgt is None:
        return colsum(x2d)
    colsum(x2d, tgt)
    _ready(p)
    return None


def _adjacent(a, b, c):
    """True when tensors a, b, c are back-to-back views of ONE storage (FlatState lays q/k/v weights out that way).
    Separate allocations that merely happen to be neighbours do not count: a view may not span storages."""
    if a is None or b is None or c is None or not (a.is_contiguous() and b.is_contiguous() and c.is_contiguous()):
        return False
    sa = a.untyped_storage()
    if sa.data_ptr() != b.untyped_storage().data_ptr() or sa.data_ptr() != c.untyped_storage().data_ptr():
        return False
    es = a.element_size()
    return (a.storage_offset() + a.numel() == b.storage_offset() and b.storage_offset() + b.numel() == c.storage_offset()
            and (c.storage_offset() + c.numel()) * es <= sa.nbytes())


def _span3(a, shape):
    """View of 3x the extent of `a` starting at a (requires _adjacent)."""
    return torch.empty(0, device=a.device, dtype=a.dtype).set_(a.untyped_storage(), a.storage_offset(), shape)


# ------------------------------------------------------------------ Linear
class LinearFn(torch.autograd.Function):
    """y = x W^T + b (eight_mile Dense / nn.Linear).  x [M, K] compute dtype."""

    @staticmethod
    def forward(ctx, x, weight, bias, out_f32):
        M, Kd = x.shape
        N = weight.shape[0]
        e = cached_weights(weight, x.dtype)
        wc, wcT = e if e is not None else (_wcast(weight, x.dtype), None)
        y = K.gemm(L.NT, x, wc, M, N, Kd, bias=bias.detach() if bias is not None else None,
                   out_dtype=torch.float32 if out_f32 else x.dtype)
        ctx.save_for_backward(x, wc)
        ctx.wcT = wcT
        ctx.has_bias = bias is not None
        ctx.params = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wc = ctx.saved_tensors
        weight, bias = ctx.params
        M, Kd = x.shape
        N = wc.shape[0]
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = K.cast(dy, x.dtype)
        dx = _dgrad(dy, wc, ctx.wcT, M, Kd, N) if ctx.needs_input_grad[0] else None
        dw = acc_wgrad(weight, dy, x, N, Kd, M) if ctx.needs_input_grad[1] else None
        db = acc_colsum(bias, dy) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db, None


def linear(x, weight, bias, out_f32=False):
    shp = x.shape
    y = LinearFn.apply(x.reshape(-1, shp[-1]), weight, bias, out_f32)
    return y.view(*shp[:-1], weight.shape[0])


# ------------------------------------------------------------------ LayerNorm / dropout / mask / gather
class LayerNormFn(torch.autograd.Function):
    """y = LN(res + dropout_p(x)); res may be None."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps, p):
        seed = next_seed() if p > 0 else 0
        y, mean, rstd = K.layernorm_fwd(x, res, gamma.detach(), beta.detach(), eps, p, seed, 1)
        ctx.save_for_backward(x, res, gamma, mean, rstd)
        ctx.p, ctx.seed = p, seed
        ctx.params = (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, res, gamma, mean, rstd = ctx.saved_tensors
        pg, pb = ctx.params
        tg, tb = _target(pg), _target(pb)
        direct = tg is not None and tb is not None
        dg = tg if direct else torch.zeros_like(gamma, dtype=torch.float32)
        db = tb if direct else torch.zeros_like(gamma, dtype=torch.float32)
        dx, dres = K.layernorm_bwd(dy.contiguous(), None, x, res, gamma.detach(), mean, rstd, dg, db, ctx.p, ctx.seed, 1,
                                   want_dx=True, want_dres=res is not None and ctx.p > 0)
        if res is not None and ctx.p == 0:
            dres = dx
        if direct:
            _ready(pg, pb)
            dg = db = None
        return dx, (dres if res is not None else None), dg, db, None, None


def layer_norm(x, gamma, beta, eps, res=None, p=0.0):
    return LayerNormFn.apply(x.contiguous(), res, gamma, beta, eps, p)


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p):
        ctx.p, ctx.seed = p, next_seed()
        return K.dropout(x.contiguous(), p, ctx.seed, 2)

    @staticmethod
    def backward(ctx, dy):
        return K.dropout(dy.contiguous(), ctx.p, ctx.seed, 2), None


def dropout(x, p, training):
    if not training or p <= 0:
        return x
    return DropoutFn.apply(x, p)


class MaskFillRowsFn(torch.autograd.Function):
    """features[time_mask] = mask_emb  (audio8/wav2vec2.py:939,717).  x [rows, d]; mask uint8 [rows]."""

    @staticmethod
    def forward(ctx, x, mask_u8, emb):
        y = K.mask_fill_rows_(x.clone(), mask_u8, emb.detach())
        ctx.save_for_backward(mask_u8)
        ctx.params = (emb,)
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask_u8,) = ctx.saved_tensors
        (emb,) = ctx.params
        tgt = _target(emb)
        demb = tgt if tgt is not None else torch.zeros((dy.shape[-1],), device=dy.device, dtype=torch.float32)
        dx = K.mask_fill_rows_bwd_(dy.clone(), mask_u8, demb)
        if tgt is not None:
            _ready(emb)
            demb = None
        return dx, None, demb


class GatherRowsFn(torch.autograd.Function):
    """x[idx] for unique row indices (masked select of frames, audio8/wav2vec2.py:946,381)."""

    @staticmethod
    def forward(ctx, x, idx_i32):
        ctx.save_for_backward(idx_i32)
        ctx.rows = x.shape[0]
        return K.gather_rows(x.contiguous(), idx_i32)

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        dx = K.zeros((ctx.rows, dy.shape[1]), dy.dtype, dy.device)
        K.scatter_add_rows_(dx, idx, dy.contiguous())
        return dx, None


class ZeroFn(torch.autograd.Function):
    """x[~keep_rows] = 0 and/or x[:, :, cmask] = 0; the same masking is its own backward."""

    @staticmethod
    def forward(ctx, x, keep_rows_u8, cmask_u8):
        ctx.save_for_backward(keep_rows_u8, cmask_u8)
        y = x.clone()
        if keep_rows_u8 is not None:
            K.zero_rows_(y.view(-1, y.shape[-1]), keep_rows_u8)
        if cmask_u8 is not None:
            K.mask_zero_cols_(y, cmask_u8)
        return y

    @staticmethod
    def backward(ctx, dy):
        keep_rows_u8, cmask_u8 = ctx.saved_tensors
        dx = dy.clone()
        if keep_rows_u8 is not None:
            K.zero_rows_(dx.view(-1, dx.shape[-1]), keep_rows_u8)
        if cmask_u8 is not None:
            K.mask_zero_cols_(dx, cmask_u8)
        return dx, None, None


# ------------------------------------------------------------------ conv feature extractor
_CONV_DX_CACHE = {}       # (utterances, padded rows, channels, dtype, device, layer) -> data-gradient buffer whose pad rows are zero


class ConvStackFn(torch.autograd.Function):
    """ConvFeatureExtractionModel.forward (audio8/wav2vec2.py:448-456) -> channels-last [B, T, 512].
    Layer 0 = fused conv+GroupNorm+GELU kernels; layers 1.. = one NT GEMM each over overlapping rows."""

    @staticmethod
    def forward(ctx, x, gn_w, gn_b, feats, *ws):
        with phase("conv_fwd"):
            return ConvStackFn._forward(ctx, x, gn_w, gn_b, feats, *ws)

    @staticmethod
    def backward(ctx, dout):
        with phase("conv_bwd"):
            return ConvStackFn._backward(ctx, dout)

    @staticmethod
    def _forward(ctx, x, gn_w, gn_b, feats, *ws):
        dtype = compute_dtype()
        B, Ls = x.shape
        nl = len(ws)
        x = x.contiguous().float()
        dev = x.device
        lens, n = [], Ls
        for co, k, s in feats:
            n = (n - k) // s + 1
            lens.append(n)
        acts = [torch.empty((B, lens[i], feats[i][0]), device=dev, dtype=dtype) for i in range(nl)]
        zs = [None] + [torch.empty((B * lens[i], feats[i][0]), device=dev, dtype=dtype) for i in range(1, nl)]
        stats = torch.empty((B, feats[0][0], 2), device=dev, dtype=torch.float32)
        wcs, wcTs = [None], [None]
        for i in range(1, nl):
            co, k, s = feats[i]
            cin = ws[i].shape[1]
            e = cached_weights(ws[i], dtype)
            if e is not None:
                wc, wcT = e
            else:
                w2 = ws[i].detach().permute(0, 2, 1).reshape(co, k * cin)        # [co][kk*cin + ci]
                wc, wcT = (w2.contiguous() if dtype == torch.float32 else K.cast(w2.contiguous(), dtype)), None
            wcs.append(wc)
            wcTs.append(wcT)
        w0c, gw, gb = ws[0].detach().contiguous(), gn_w.detach(), gn_b.detach()

        def run_layers(b0, b1, lo, hi):                # layers lo .. hi-1 on utterances [b0, b1)
            for i in range(lo, hi):
                if i == 0:
                    with phase("conv0_fwd"):
                        K.conv0_gn_gelu_fwd(x[b0:b1], w0c, gw, gb, dtype, out=acts[0][b0:b1], stats=stats[b0:b1])
                    continue
                co, k, s = feats[i]
                cin, Lin, Lout = feats[i - 1][0], lens[i - 1], lens[i]
                K.gemm(L.NT, acts[i - 1][b0:b1], wcs[i], (b1 - b0) * Lout, co, k * cin, lda=s * cin, a_rpb=Lout, a_bs=Lin * cin, act=L.ACT_GELU,
                       zout=zs[i][b0 * Lout:b1 * Lout], out=acts[i][b0:b1].view(-1, co))

        ch, nlc = _conv_chunking(B, nl)
        for b0 in range(0, B, ch):                     # the big leading layers utterance-chunk by utterance-chunk: each layer's
            run_layers(b0, min(B, b0 + ch), 0, nlc)    # output is consumed by the next one while it still sits in the Infinity Cache
        run_layers(0, B, nlc, nl)
        ctx.feats, ctx.lens, ctx.B = feats, lens, B
        ctx.wcTs = wcTs
        ctx.ws = ws
        ctx.params = (ws[0], gn_w, gn_b)
        ctx.save_for_backward(x, gn_w, gn_b, stats, ws[0], *acts[:-1], *zs[1:], *wcs[1:])
        ctx.nl = nl
        return acts[-1]

    @staticmethod
    def _backward(ctx, dout):
        nl, B, feats, lens = ctx.nl, ctx.B, ctx.feats, ctx.lens
        sv = ctx.saved_tensors
        x, gn_w, gn_b, stats, w0 = sv[:5]
        acts = sv[5:5 + nl - 1]                      # a_0 .. a_{nl-2}
        zs = (None,) + sv[5 + nl - 1:5 + 2 * (nl - 1)]
        wcs = (None,) + sv[5 + 2 * (nl - 1):]
        dtype = acts[0].dtype
        grads = [None] * nl
        da = dout.contiguous()
        if da.dtype != dtype:
            da = K.cast(da, dtype)
        co = feats[-1][0]
        dz = K.gelu_bwd_bias(da.view(-1, co), zs[nl - 1], None)
        ws = ctx.ws
        fusable = (CONV_DGRAD_CACHE is not None and WEIGHT_CACHE_DTYPE == dtype and feats[-1][1] == 2 and all(f[2] == 2 and f[1] in (2, 3) for f in feats[1:])
                   and all(ctx.wcTs[i] is not None for i in range(1, nl)) and all(feats[i][1] == 2 or id(ws[i]) in CONV_DGRAD_CACHE for i in range(1, nl)))

        def bwd_range(b0, b1, dz, pad, hi, lo, last):
            """Layers hi .. lo (descending, lo >= 1) for utterances [b0, b1): dz = gradient w.r.t. layer hi's pre-activation, stored with
            `pad` zero rows around each utterance.  Returns the same pair for layer lo - 1."""
            nb = b1 - b0
            for i in range(hi, lo - 1, -1):
                co, k, s = feats[i]
                cin = acts[i - 1].shape[-1]
                Lin, Lout = lens[i - 1], lens[i]
                M = nb * Lout
                dzrows = Lout + 2 * pad
                akw = dict(a_off=pad * co, a_rpb=Lout, a_bs=dzrows * co) if pad else {}
                a_prev = acts[i - 1][b0:b1]
                tgt = _target(ws[i])
                deferred = None
                if tgt is not None:
                    # off the critical path: wgrad GEMM + re-layout [co][k][cin] -> [co][cin][k] + accumulate on the side stream
                    def run(dz=dz, a=a_prev, co=co, k=k, cin=cin, M=M, s=s, Lout=Lout, Lin=Lin, akw=akw, tgt=tgt, p=ws[i]):
                        # split-K partial sums land in a buffer kept across steps (the fold kernel clears it behind its read:
                        # no fill launch), then fold into the parameter's [co][cin][k] layout
                        key = (co, k, cin, dz.device)
                        dw2 = _CONV_DW_CACHE.get(key)
                        if dw2 is None:
                            dw2 = _CONV_DW_CACHE[key] = torch.zeros((co, k * cin), device=dz.device, dtype=torch.float32)
                        K.gemm(L.TN, dz, a, co, k * cin, M, out=dw2, splitk=_splitk(_tiles(co, k * cin), M), ldb=s * cin, b_rpb=Lout,
                               b_bs=Lin * cin, lda=co, **akw)
                        K.conv_wgrad_fold_(dw2, tgt, co, cin, k)
                        if last:                       # the gradient is complete once the last chunk's share is enqueued
                            _ready(p)
                    if fusable and (CONV_WGRAD_DEFER == 2 or (CONV_WGRAD_DEFER == 1 and i == 1)):
                        deferred = (run, dz, a_prev)   # handed to the side stream BEHIND this layer's data gradient (see CONV_WGRAD_DEFER)
                    else:
                        _on_side(run, dz, a_prev)
                else:
                    dw2 = wgrad(dz, a_prev, co, k * cin, M, ldb=s * cin, b_rpb=Lout, b_bs=Lin * cin, lda=co, **akw)
                    gi = dw2.view(co, k, cin).permute(0, 2, 1).contiguous()
                    grads[i] = gi if grads[i] is None else grads[i] + gi
                zprev = zs[i - 1][b0 * Lin:b1 * Lin] if i - 1 >= 1 else None
                if not fusable:
                    dcol = _dgrad(dz, wcs[i], ctx.wcTs[i], M, k * cin, co)
                    dz = K.col2im_gelu_grad(dcol, zprev, nb, Lin, Lout, cin, k, s).view(-1, cin)
                    del dcol
                    continue
                # fused data gradient: write dx[b, p, :] (* gelu'(z)) straight from the GEMM epilogues
                npad = 1 if (i - 1 >= 1 and feats[i - 1][1] == 3) else 0          # the layer below reads dz[t-1], dz[t]
                prow = Lin + 2 * npad
                # The GEMMs below write positions 0 .. 2 Lout (- 1) of every utterance and nothing else: the pad rows and the
                # uncovered tail of a buffer that was zero-filled ONCE stay zero, so the buffers are kept across steps (8 small
                # fill launches per step less on the critical stream; 0.8 GB resident at cfg2).
                key = (nb, prow, cin, dtype, dz.device, i)
                dx = _CONV_DX_CACHE.get(key)
                if dx is None:
                    dx = _CONV_DX_CACHE[key] = torch.zeros((nb * prow, cin), device=dz.device, dtype=dtype)
                act = L.ACT_MUL_GELU_GRAD if zprev is not None else L.ACT_NONE
                zk = dict(zin=zprev, ldz=2 * cin, z_bs=Lin * cin) if zprev is not None else {}
                if k == 2:
                    K.gemm(L.NT, dz, ctx.wcTs[i], M, 2 * cin, co, out=dx, lda=co, ldc=2 * cin, c_off=npad * cin, c_rpb=Lout, c_bs=prow * cin,
                           act=act, **akw, **(dict(zk, z_rpb=Lout) if zk else {}))
                else:
                    assert pad == 1
                    b_even, b_odd = CONV_DGRAD_CACHE[id(ws[i])]
                    # even positions p = 2t, t = 0..Lout: rows [dz[t-1] | dz[t]] of the zero-padded dz (K = 2 co)
                    even = (dz, b_even, nb * (Lout + 1), cin, 2 * co,
                            dict(out=dx, lda=co, a_rpb=Lout + 1, a_bs=dzrows * co, ldc=2 * cin, c_off=npad * cin, c_rpb=Lout + 1, c_bs=prow * cin,
                                 act=act, **(dict(zk, z_rpb=Lout + 1) if zk else {})))
                    # odd positions p = 2t + 1, t = 0..Lout-1
                    odd = (dz, b_odd, M, cin, co,
                           dict(out=dx, lda=co, a_off=co, a_rpb=Lout, a_bs=dzrows * co, ldc=2 * cin, c_off=(npad + 1) * cin, c_rpb=Lout, c_bs=prow * cin,
                                act=act, **(dict(zk, z_rpb=Lout, z_off=cin) if zk else {})))
                    if CONV_DGRAD_GROUPED:               # one grid for both halves: no tail / ramp between two half-size launches
                        K.gemm_grouped(L.NT, [even, odd])
                    else:
                        for A_, B_, m_, n_, k_, kw_ in (even, odd):
                            K.gemm(L.NT, A_, B_, m_, n_, k_, **kw_)
                if deferred is not None:
                    _on_side(*deferred)
                dz, pad = dx, npad
            return dz, pad

        pw0, pgw, pgb = ctx.params
        t0, t1, t2 = _target(pw0), _target(pgw), _target(pgb)
        direct = t0 is not None and t1 is not None and t2 is not None
        dw0 = t0 if direct else torch.zeros_like(w0, dtype=torch.float32)
        dgw = t1 if direct else torch.zeros_like(gn_w, dtype=torch.float32)
        dgb = t2 if direct else torch.zeros_like(gn_b, dtype=torch.float32)
        w0c, gw, gb = w0.detach().contiguous(), gn_w.detach(), gn_b.detach()
        ch, nlc = _conv_chunking(B, nl)
        nlc = max(nlc, 1)
        dz, pad = bwd_range(0, B, dz, 0, nl - 1, nlc, True)          # the small trailing layers on the whole batch
        rows = dz.shape[0] // B                                       # rows per utterance of layer (nlc - 1)'s gradient, pad included
        dzv = dz.view(B, rows, -1)
        for b0 in range(0, B, ch):
            b1 = min(B, b0 + ch)
            dzc, padc = bwd_range(b0, b1, dzv[b0:b1].reshape(-1, dzv.shape[-1]), pad, nlc - 1, 1, b1 == B)
            assert padc == 0
            with phase("conv0_bwd"):
                K.conv0_gn_gelu_bwd(x[b0:b1], w0c, gw, gb, stats[b0:b1], dzc.view(b1 - b0, lens[0], -1), dw0, dgw, dgb)
        if direct:
            _ready(pw0, pgw, pgb)
            dw0 = dgw = dgb = None
        grads[0] = dw0
        return (None, dgw, dgb, None, *grads)


_CONV_DW_CACHE = {}        # (co, k, cin, device) -> f32 [co, k*cin] split-K landing buffer of a conv weight gradient, zero between uses


# Encoder forward as two half-batches on two streams (layers.TransformerEncoderStack sets HALF_STREAM around its layer loop when
# HALF_BATCH_STREAMS is on; a8_encoder_layer_fwd then runs once per half).  The dropout counters carry each half's origin in the
# logical batch, so the masks are those of the whole-batch launch.  With HALF_BATCH_BWD the layers' backward chains split the same way:
# HalfEdgeFn marks the two ends of the stack in the autograd graph, its backward at the top lets aux wait for the incoming gradient, the
# one at the bottom makes the caller's stream wait for aux; a layer that cannot split (see EncoderLayerFn._backward) joins first.
HALF_BATCH_STREAMS = True
HALF_BATCH_BWD = True
LN_PARTIALS = True        # LayerNorm-backward column sums as per-block partial rows, reduced on the weight-gradient stream (-0.06 ms per step)
HALF_STREAM = None
_BWD_AUX_LIVE = None      # the aux stream while second-half gradients are in flight on it (autograd thread only)


def _bwd_fork(aux):
    """aux will run second-half backward work: order it behind the gradients enqueued so far on the current stream (once)."""
    global _BWD_AUX_LIVE
    if _BWD_AUX_LIVE is None:
        aux.wait_stream(torch.cuda.current_stream())
        _BWD_AUX_LIVE = aux


def _bwd_join():
    """Whole-batch work follows on the current stream: wait for the second halves."""
    global _BWD_AUX_LIVE
    if _BWD_AUX_LIVE is not None:
        torch.cuda.current_stream().wait_stream(_BWD_AUX_LIVE)
        _BWD_AUX_LIVE = None


class HalfEdgeFn(torch.autograd.Function):
    """Identity at the two ends of the encoder stack; its backward forks (top) or joins (bottom) the half-batch backward streams."""

    @staticmethod
    def forward(ctx, x, aux, top):
        ctx.aux, ctx.top = aux, top
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        global _BWD_AUX_LIVE
        if ctx.top:
            _BWD_AUX_LIVE = None                      # (a backward pass that died half-way may have left it set)
            if HALF_BATCH_BWD and WGRAD_SIDE_STREAM is not None:
                dy = dy.contiguous()
                dy.record_stream(ctx.aux)
                _bwd_fork(ctx.aux)
        else:
            _bwd_join()
        return dy, None, None
_HALF_AUX = {}


def half_stream(device):
    """The auxiliary stream of the half-batch forward (one per device), or None when the feature is off."""
    if not HALF_BATCH_STREAMS or not COMPOSITE_ENCODER_FWD or K.GEMM_PROFILE is not None:
        return None
    s = _HALF_AUX.get(device)
    if s is None:
        s = _HALF_AUX[device] = torch.cuda.Stream(device=device)
    return s


# Utterance chunking of the conv stack's big leading layers (layer 0's output is 51 MB per utterance at 250 k samples): with
# CONV_CHUNK_UTTS utterances per chunk, layers 0 .. CONV_CHUNK_LAYERS-1 run chunk by chunk so that a layer reads its input
# while it is still resident in the 256 MB Infinity Cache; the remaining (small) layers run on the whole batch.  0 = off.
CONV_CHUNK_UTTS = 0
CONV_CHUNK_LAYERS = 3
# even / odd halves of a k = 3, stride 2 conv data gradient as ONE grouped launch (A/B knob)
CONV_DGRAD_GROUPED = True
# When a conv layer's weight gradient is handed to the side stream: 0 = ahead of the layer's data gradient (both MFMA-bound, they share the chip);
# 1 = layer 1's BEHIND its data gradient, so that it runs beside conv layer 0's backward (VALU-bound, the matrix cores idle: the last 0.25 ms of
# backward had nothing beside it); 2 = every layer's behind its own data gradient.
CONV_WGRAD_DEFER = 0


def _conv_chunking(B, nl):
    ch = CONV_CHUNK_UTTS
    if ch <= 0 or ch >= B:
        return B, 0
    return ch, max(1, min(CONV_CHUNK_LAYERS, nl - 1))


# ------------------------------------------------------------------ positional conv
class PosConvFn(torch.autograd.Function):
    """y = x + gelu(conv1d_same(x; w) + bias), grouped, k = 128, with w = weight_norm(v, g; dim=2)
    (audio8/wav2vec2.py:609, 634-638).  v f32 [d, d/G, k], g f32 [1, 1, k]: the weight norm and its gradient are
    part of this node (a8_weightnorm_pack / a8_weightnorm_bwd), not of the tape."""

    @staticmethod
    def forward(ctx, *args):
        with phase("posconv_fwd"):
            return PosConvFn._forward(ctx, *args)

    @staticmethod
    def backward(ctx, dy):
        with phase("posconv_bwd"):
            return PosConvFn._backward(ctx, dy)

    @staticmethod
    def _forward(ctx, x, v, g, bias, groups, pad_left):
        B, T, d = x.shape
        k = v.shape[2]
        wf, wt, inv_norm = K.weightnorm_pack(v.detach(), g.detach(), groups, x.dtype)
        y, z = K.posconv_fwd(x.contiguous(), wf, bias.detach(), groups, k, pad_left, 1, want_z=True)
        ctx.save_for_backward(x, z, wt, inv_norm)
        ctx.groups, ctx.pad_left, ctx.k = groups, pad_left, k
        ctx.params = (v, g, bias)
        return y

    @staticmethod
    def _backward(ctx, dy):
        x, z, wt, inv_norm = ctx.saved_tensors
        B, T, d = x.shape
        G, k = ctx.groups, ctx.k
        dg = d // G
        dy = dy.contiguous()
        pv, pg, pbias = ctx.params
        tb = _target(pbias)
        dbias = tb if tb is not None else torch.zeros((d,), device=x.device, dtype=torch.float32)
        dz = K.gelu_bwd_bias(dy, z, dbias)
        if tb is not None:
            _ready(pbias)
            dbias = None
        dxc, _ = K.posconv_fwd(dz, wt, None, G, k, k - 1 - ctx.pad_left, -1)
        dx = K.add(dy, dxc)
        tv, tg = _target(pv), _target(pg)
        direct = tv is not None and tg is not None
        dv = tv if direct else torch.zeros(pv.shape, device=x.device, dtype=torch.float32)
        dgain = tg if direct else torch.zeros(pg.shape, device=x.device, dtype=torch.float32)

        def wgrad_chain():
            dwg = K.zeros((G, dg, k, dg), torch.float32, x.device)
            K.posconv_wgrad(x, dz, dwg, G, k, ctx.pad_left)
            K.weightnorm_bwd_(dwg, pv.detach(), pg.detach(), inv_norm, dv, dgain, G)
            if direct:
                _ready(pv, pg)
        if direct and SIDE_EXTRAS:               # feeds nothing downstream: beside the data-gradient chain, on the side stream
            _on_side(wgrad_chain, x, dz, inv_norm)
            dv = dgain = None
        else:
            wgrad_chain()
            if direct:
                dv = dgain = None
        return dx, dv, dgain, dbias, None, None


# ------------------------------------------------------------------ transformer encoder layer (post-LN)
# q/k/v bias gradients: a column-sum pass over dqkv (default), or accumulated by the attention backward kernels themselves
# (a8_attn_bwd's dbias: block-level column sums, one atomic per column per block).  Same-box A/B: 13.67-13.75 ms fused vs
# 13.64-13.69: the extra barriers and 2.4e5 atomics cost a little more than the 12 small passes they replace.
ATTN_BIAS_FUSED = False
COMPOSITE_ENCODER_FWD = True          # one C call per encoder layer forward (host time; the kernels are the same)
COMPOSITE_ENCODER_BWD = True          # ... and per data-gradient chain of its backward
COMPOSITE_ENCODER_BWD_FULL = True     # ... or (round 6) ONE call for the whole layer's backward incl. its side-stream work (a8_encoder_layer_bwd_full)


class EncoderLayerFn(torch.autograd.Function):
    """One eight_mile TransformerEncoder with layer_norms_after=True:
        h = W_O attn(W_Q x, W_K x, W_V x);  x1 = ln2(x + drop(h));  x2 = ln1(x1 + drop(W_2 gelu(W_1 x1)))
    12 kernel launches forward, 21 backward; QKV run as one N = 3d GEMM on a concatenated working copy."""

    @staticmethod
    def forward(ctx, *args):
        with phase("enc_fwd"):
            return EncoderLayerFn._forward(ctx, *args)

    @staticmethod
    def backward(ctx, dx2):
        if HOST_DELAY_US:                 # developer probe (tools/step_ab.py hostdelay=<us>): is the launching thread on the critical path?
            t_end = time.perf_counter() + HOST_DELAY_US * 1e-6
            while time.perf_counter() < t_end:
                pass
        with phase("enc_bwd"):
            return EncoderLayerFn._backward(ctx, dx2)

    @staticmethod
    def _forward(ctx, x, keymask, B, T, H, p, eps, wq, bq, wk, bk, wv, bv, wo, bo, w1, b1, w2, b2, g1, be1, g2, be2):
        M, d = x.shape
        dt = x.dtype
        dff = w1.shape[0]
        fused = _adjacent(wq, wk, wv) and _adjacent(bq, bk, bv)      # FlatState layout: no concatenation needed
        bqkv = _span3(bq.detach(), (3 * d,)) if fused else torch.cat([bq.detach(), bk.detach(), bv.detach()], 0)
        eq = cached_weights(wq, dt) if fused else None
        if eq is not None:
            wqkv_c, wqkv_t = eq
        else:
            wqkv = _span3(wq.detach(), (3 * d, d)) if fused else torch.cat([wq.detach(), wk.detach(), wv.detach()], 0)
            wqkv_c, wqkv_t = _wcast(wqkv, dt), None
        eo, e1, e2 = cached_weights(wo, dt), cached_weights(w1, dt), cached_weights(w2, dt)
        wo_c, wo_t = eo if eo is not None else (_wcast(wo, dt), None)
        w1_c, w1_t = e1 if e1 is not None else (_wcast(w1, dt), None)
        w2_c, w2_t = e2 if e2 is not None else (_wcast(w2, dt), None)
        if HALF_STREAM is not None and (eq is None or eo is None or e1 is None or e2 is None):
            # no prepared weight copies (no Stepper): the casts above were enqueued on this stream just now, and the half-batch stream
            # only met this one at the top of the stack
            HALF_STREAM.wait_stream(torch.cuda.current_stream())
            # ... and they are temporaries: under no_grad nothing keeps them alive past this call, so without a note to the allocator
            # the next layer's casts could be handed their memory while the half-batch stream still reads it (round 6: the bf16 eval-mode
            # acoustic model without a Stepper decoded differently from run to run, tests/test_parity_hardening_gpu.py)
            for w_ in (wqkv_c, wo_c, w1_c, w2_c, bqkv):
                if w_ is not None and w_.is_cuda:
                    w_.record_stream(HALF_STREAM)
        if COMPOSITE_ENCODER_FWD and K.GEMM_PROFILE is None and x.is_contiguous():
            # the whole launch sequence below behind one C call (a8_encoder_layer_fwd); the per-launch path stays for the instrumented
            # roofline steps of bench.py (per-GEMM events) and as the readable statement of the sequence
            seed_a, seed1, seed2 = (next_seed(), next_seed(), next_seed()) if p > 0 else (0, 0, 0)
            x2, qkv, o, lse, ao, x1, mean2, rstd2, z1, h, f, mean1, rstd1 = K.encoder_layer_fwd(
                x, keymask, B, T, H, p, eps, (seed_a, seed1, seed2), wqkv_c, bqkv, wo_c, bo, w1_c, b1, w2_c, b2, g1, be1, g2, be2,
                aux_stream=HALF_STREAM)
            ctx.save_for_backward(x, keymask, qkv, o, lse, ao, x1, mean2, rstd2, z1, h, f, mean1, rstd1,
                                  wqkv_c, wo_c, w1_c, w2_c, g1, g2)
            ctx.cfg = (B, T, H, p, seed_a, seed1, seed2)
            ctx.wts = (wqkv_t, wo_t, w1_t, w2_t)
            ctx.params = (wq, bq, wk, bk, wv, bv, wo, bo, w1, b1, w2, b2, g1, be1, g2, be2)
            ctx.half_aux = HALF_STREAM if B >= 2 else None
            return x2
        ctx.half_aux = None
        qkv = K.gemm(L.NT, x, wqkv_c, M, 3 * d, d, bias=bqkv)
        seed_a = next_seed() if p > 0 else 0
        o, lse = K.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, T, H, 3 * d, 1.0 / math.sqrt(d // H), keymask, p, seed_a)
        ao = K.gemm(L.NT, o, wo_c, M, d, d, bias=bo.detach())
        seed1 = next_seed() if p > 0 else 0
        x1, mean2, rstd2 = K.layernorm_fwd(ao, x, g2.detach(), be2.detach(), eps, p, seed1, 3)
        z1 = torch.empty((M, dff), device=x.device, dtype=dt)
        h = K.gemm(L.NT, x1, w1_c, M, dff, d, bias=b1.detach(), act=L.ACT_GELU, zout=z1)
        f = K.gemm(L.NT, h, w2_c, M, d, dff, bias=b2.detach())
        seed2 = next_seed() if p > 0 else 0
        x2, mean1, rstd1 = K.layernorm_fwd(f, x1, g1.detach(), be1.detach(), eps, p, seed2, 4)
        ctx.save_for_backward(x, keymask, qkv, o, lse, ao, x1, mean2, rstd2, z1, h, f, mean1, rstd1,
                              wqkv_c, wo_c, w1_c, w2_c, g1, g2)
        ctx.cfg = (B, T, H, p, seed_a, seed1, seed2)
        ctx.wts = (wqkv_t, wo_t, w1_t, w2_t)
        ctx.params = (wq, bq, wk, bk, wv, bv, wo, bo, w1, b1, w2, b2, g1, be1, g2, be2)
        return x2

    @staticmethod
    def _backward(ctx, dx2):
        (x, keymask, qkv, o, lse, ao, x1, mean2, rstd2, z1, h, f, mean1, rstd1, wqkv_c, wo_c, w1_c, w2_c, g1, g2) = ctx.saved_tensors
        B, T, H, p, seed_a, seed1, seed2 = ctx.cfg
        M, d = x.shape
        dff = z1.shape[1]
        dev = x.device
        f32 = torch.float32
        dx2 = dx2.contiguous()
        wq, bq, wk, bk, wv, bv, wo, bo, w1, b1, w2, b2, pg1, pbe1, pg2, pbe2 = ctx.params

        def ln_targets(pg, pb):
            tg, tb = _target(pg), _target(pb)
            if tg is not None and tb is not None:
                return tg, tb, True
            return torch.zeros(d, device=dev, dtype=f32), torch.zeros(d, device=dev, dtype=f32), False

        dg1, dbe1, dir1 = ln_targets(pg1, pbe1)
        dg2, dbe2, dir2 = ln_targets(pg2, pbe2)
        # ln1: x2 = LN(x1 + drop(f))
        tb2, tbo = _target(b2), _target(bo)      # bias gradients of w_2 / w_O ride on the LayerNorm backward passes
        wqkv_t, wo_t, w1_t, w2_t = ctx.wts
        tb1 = _target(b1)
        tw, tb = [_target(w) for w in (wq, wk, wv)], [_target(b) for b in (bq, bk, bv)]
        direct_qkv = all(t is not None for t in tw + tb) and _adjacent(*tw) and _adjacent(*tb)
        if (COMPOSITE_ENCODER_BWD and K.GEMM_PROFILE is None and GROUP_WGRAD and dir1 and dir2 and direct_qkv and not ATTN_BIAS_FUSED
                and all(t is not None for t in (tb2, tbo, tb1, wqkv_t, wo_t, w1_t, w2_t, _target(w1), _target(w2), _target(wo)))):
            # the data-gradient chain up to the attention core behind one C call (a8_encoder_layer_bwd); the statement of the sequence,
            # and every other case (no direct gradient targets, no prepared transposes, instrumented steps), is the code below
            aux = _BWD_AUX_LIVE if (_BWD_AUX_LIVE is not None and _BWD_AUX_LIVE is ctx.half_aux) else None
            if aux is None:
                _bwd_join()
            side = WGRAD_SIDE_STREAM
            if (COMPOSITE_ENCODER_BWD_FULL and side is not None and LN_PARTIALS and SIDE_EXTRAS
                    and all(t_.is_contiguous() for t_ in (x, qkv, o, ao, x1, z1, h, f))):
                # the whole layer behind ONE C call (a8_encoder_layer_bwd_full): the chains, the side stream's grouped weight gradients and
                # column-sum work behind both, the QKV data gradients.  Same launches, same order per stream as the code below.
                params4 = ((w2,), (w1,), (wo,), (wq, wk, wv))
                acc = [not _first_touch(*ps) for ps in params4]              # (decided before anything is reported ready)
                dx, temps = K.encoder_layer_bwd_full(
                    dx2, x, keymask, qkv, o, lse, ao, x1, mean2, rstd2, z1, h, f, mean1, rstd1, g1, g2, w2_t, w1_t, wo_t, wqkv_t,
                    dg1, dbe1, dg2, dbe2, tb2, tb1, tbo, _target(w2), _target(w1), _target(wo), _span3(tw[0], (3 * d, d)), acc,
                    _span3(tb[0], (3 * d,)), B, T, H, p, (seed_a, seed1, seed2), aux, side)
                for t_ in (x, qkv, o, ao, x1, z1, h, f, lse):               # the saved forward tensors are read on the side stream too
                    t_.record_stream(side)
                with torch.cuda.stream(side):                                # every parameter gradient of the layer is final in side-stream order
                    _ready(w2, w1, wo, wq, wk, wv, pg1, pbe1, b2, b1, pg2, pbe2, bo, bq, bk, bv)
                return (dx,) + (None,) * 22
            # the LayerNorm / bias column sums leave the chain as per-block partial rows and are added up on the weight-gradient stream
            # (no same-address atomics on the critical streams: LN_PARTIALS)
            lnp = LN_PARTIALS and WGRAD_SIDE_STREAM is not None
            df, dz1, dao, dqkv, dx_res, lnws = K.encoder_layer_bwd(dx2, x, keymask, qkv, o, lse, ao, x1, mean2, rstd2, z1, f, mean1, rstd1, g1, g2,
                                                                   w2_t, w1_t, wo_t, dg1, dbe1, dg2, dbe2, tb2, tb1, tbo, B, T, H, p,
                                                                   (seed_a, seed1, seed2), aux_stream=aux, ln_partials=lnp)
            wg = WgradGroup()
            if lnws is not None:
                _ready(b1)

                def ln_colsums(ws=lnws, t1=(dg1, dbe1, tb2), t2=(dg2, dbe2, tbo), ps=(pg1, pbe1, b2, pg2, pbe2, bo)):
                    K.colsum_partials_reduce(ws[0], *t1)
                    K.colsum_partials_reduce(ws[1], *t2)
                    _ready(*ps)
                wg.extra.append(ln_colsums)
                wg.extra_inputs.append(lnws)
            else:
                _ready(pg1, pbe1, b2, b1, pg2, pbe2, bo)
            wg.add((w2,), _target(w2), df, h, d, dff, M)
            wg.add((w1,), _target(w1), dz1, x1, dff, d, M)
            wg.add((wo,), _target(wo), dao, o, d, d, M)
            bspan = _span3(tb[0], (3 * d,))
            if SIDE_EXTRAS:
                def bias_colsum(dqkv=dqkv, bspan=bspan, ps=(bq, bk, bv)):
                    colsum(dqkv, bspan)
                    _ready(*ps)
                wg.extra.append(bias_colsum)
            else:
                colsum(dqkv, bspan)
                _ready(bq, bk, bv)
            wg.add((wq, wk, wv), _span3(tw[0], (3 * d, d)), dqkv, x, 3 * d, d, M)
            wg.flush()
            if aux is None:
                dx = _dgrad(dqkv, wqkv_c, wqkv_t, M, d, 3 * d, act=L.ACT_ADD, zin=dx_res, ldz=d)
            else:
                dx = torch.empty((M, d), device=dev, dtype=x.dtype)
                dx.record_stream(aux)
                ra = (B // 2) * T
                _dgrad(dqkv[:ra], wqkv_c, wqkv_t, ra, d, 3 * d, act=L.ACT_ADD, zin=dx_res[:ra], ldz=d, out=dx[:ra])
                with torch.cuda.stream(aux):
                    _dgrad(dqkv[ra:], wqkv_c, wqkv_t, M - ra, d, 3 * d, act=L.ACT_ADD, zin=dx_res[ra:], ldz=d, out=dx[ra:])
            return (dx,) + (None,) * 22
        _bwd_join()
        df, dx1_res = K.layernorm_bwd(dx2, None, f, x1, g1.detach(), mean1, rstd1, dg1, dbe1, p, seed2, 4, want_dx=True, want_dres=p > 0, dbias=tb2)
        if p == 0:
            dx1_res = df
        if dir1:
            _ready(pg1, pbe1)
            dg1 = dbe1 = None
        # FFN
        wg = WgradGroup() if GROUP_WGRAD else None
        if tb2 is not None:
            _ready(b2)
            db2 = None
        else:
            db2 = acc_colsum(b2, df)
        dw2 = acc_wgrad_grouped(wg, w2, df, h, d, dff, M)
        wqkv_t, wo_t, w1_t, w2_t = ctx.wts
        tb1 = _target(b1)                       # b_1's gradient = column sums of dz1: accumulated by the same epilogue
        dz1 = _dgrad(df, w2_c, w2_t, M, dff, d, act=L.ACT_MUL_GELU_GRAD, zin=z1, ldz=dff, colsum=tb1)
        if tb1 is not None:
            _ready(b1)
            db1 = None
        else:
            db1 = acc_colsum(b1, dz1)
        dw1 = acc_wgrad_grouped(wg, w1, dz1, x1, dff, d, M)
        dx1_ffn = _dgrad(dz1, w1_c, w1_t, M, d, dff)
        # ln2: x1 = LN(x + drop(ao))
        dao, dx_res = K.layernorm_bwd(dx1_res, dx1_ffn, ao, x, g2.detach(), mean2, rstd2, dg2, dbe2, p, seed1, 3, want_dx=True, want_dres=p > 0, dbias=tbo)
        if p == 0:
            dx_res = dao
        if dir2:
            _ready(pg2, pbe2)
            dg2 = dbe2 = None
        # output projection
        if tbo is not None:
            _ready(bo)
            dbo = None
        else:
            dbo = acc_colsum(bo, dao)
        dwo = acc_wgrad_grouped(wg, wo, dao, o, d, d, M)
        do = _dgrad(dao, wo_c, wo_t, M, d, d)
        # attention core
        tw, tb = [_target(w) for w in (wq, wk, wv)], [_target(b) for b in (bq, bk, bv)]
        direct_qkv = all(t is not None for t in tw + tb) and _adjacent(*tw) and _adjacent(*tb)
        # the q / k / v bias gradients (column sums of dqkv) are accumulated by the attention backward kernels themselves
        _, _, _, dqkv = K.attn_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, do, lse, B, T, H, 3 * d, 1.0 / math.sqrt(d // H),
                                   keymask, p, seed_a, dbias=_span3(tb[0], (3 * d,)) if (direct_qkv and ATTN_BIAS_FUSED) else None)
        if direct_qkv:
            bspan = _span3(tb[0], (3 * d,))
            if ATTN_BIAS_FUSED:
                _ready(bq, bk, bv)
            elif wg is not None and SIDE_EXTRAS:     # off the critical path: with the grouped weight gradients on the side stream
                def bias_colsum(dqkv=dqkv, bspan=bspan, ps=(bq, bk, bv)):
                    colsum(dqkv, bspan)
                    _ready(*ps)
                wg.extra.append(bias_colsum)
            else:
                colsum(dqkv, bspan)
                _ready(bq, bk, bv)
            wspan = _span3(tw[0], (3 * d, d))
            if wg is not None:
                wg.add((wq, wk, wv), wspan, dqkv, x, 3 * d, d, M)
            else:
                def run_qkv():
                    K.gemm(L.TN, dqkv, x, 3 * d, d, M, out=wspan, accumulate=True, splitk=_splitk(_tiles(3 * d, d), M))
                    _ready(wq, wk, wv)
                _on_side(run_qkv, dqkv, x)
            gq = (None,) * 6
        else:
            dbqkv = colsum(dqkv)
            dwqkv = wgrad(dqkv, x, 3 * d, d, M)
            gq = (dwqkv[:d], dbqkv[:d], dwqkv[d:2 * d], dbqkv[d:2 * d], dwqkv[2 * d:], dbqkv[2 * d:])
        if wg is not None:
            wg.flush()                 # all four operand pairs are enqueued; the side stream waits for exactly that
        dx = _dgrad(dqkv, wqkv_c, wqkv_t, M, d, 3 * d, act=L.ACT_ADD, zin=dx_res, ldz=d)     # + the residual branch's gradient
        return (dx, None, None, None, None, None, None, *gq, dwo, dbo, dw1, db1, dw2, db2, dg1, dbe1, dg2, dbe2)


# ------------------------------------------------------------------ Gumbel vector quantiser
class VQFn(torch.autograd.Function):
    """GumbelVectorQuantizer.forward (audio8/wav2vec2.py:547-576): returns (q [N, G*D], prob_ppl [1]).
    The 512->G*V projection and its gradients run in the compute dtype like every other Linear (f32 logits out of the bf16
    MFMA in bf16 mode: the exact-f32 GEMM path is a latency-bound 0.3 ms per step for 2 GFLOP); selection, softmax statistics
    and perplexity are f32.  `gumbel` injects the noise (None: drawn with torch's recipe)."""

    @staticmethod
    def forward(ctx, y, wp, bp, vars_, G, V, tau, training, gumbel):
        N, Cin = y.shape
        D = vars_.shape[-1]
        y = y.contiguous()
        e = cached_weights(wp, y.dtype)
        wpc, wpt = e if e is not None else (_wcast(wp, y.dtype).contiguous(), None)
        logits = K.gemm(L.NT, y, wpc, N, G * V, Cin, bias=bp.detach(), out_dtype=torch.float32).view(N * G, V)
        vars2d = vars_.detach().reshape(G * V, D)
        if training and gumbel is None:        # the noise of F.gumbel_softmax, drawn inside the select kernel (seeded from torch's seed)
            gumbel = torch.empty_like(logits)
            codes, avg = K.vq_select(logits, None, tau, gumbel_gen=gumbel, seed=next_seed())
        else:
            codes, avg = K.vq_select(logits, gumbel if training else None, tau)
        q = K.vq_gather(vars2d, codes, N, G, V, D, y.dtype)
        ppl, dppl = K.vq_ppl(avg)
        ctx.save_for_backward(y, wpc, logits, gumbel if training else None, codes, vars2d, dppl)
        ctx.cfg = (G, V, D, tau, training, vars_.shape)
        ev = cached_weights(vars_, y.dtype)
        ctx.wts = (wpt, ev[0] if ev is not None else None)
        ctx.params = (wp, bp, vars_)
        ctx.mark_non_differentiable(codes)
        ctx.set_materialize_grads(False)         # (backward tests dppl_out for None; no zero-filled gradient of the codes)
        return q, ppl, codes

    @staticmethod
    def backward(ctx, dq, dppl_out, _dcodes):
        y, wpc, logits, gumbel, codes, vars2d, dppl = ctx.saved_tensors
        G, V, D, tau, training, vshape = ctx.cfg
        N, Cin = y.shape
        dev, dt = y.device, y.dtype
        if dq is None:                            # (only the perplexity was used downstream)
            dq = K.zeros((N, G * D), dt, dev)
        dqc = dq.contiguous()
        if dqc.dtype != dt:
            dqc = K.cast(dqc, dt)
        eproj = None
        if training:
            eproj = torch.empty((N, G * V), device=dev, dtype=torch.float32)
            vars_c = ctx.wts[1] if ctx.wts[1] is not None else (vars2d if dt == torch.float32 else K.cast(vars2d.contiguous(), dt))
            for g in range(G):
                K.gemm(L.NT, dqc, vars_c[g * V:], N, V, D, lda=G * D, a_off=g * D, out=eproj, c_off=g * V, ldc=G * V)
        pwp, pbp, pvars = ctx.params
        tv = _target(pvars)
        dvars = tv.view(G * V, D) if tv is not None else torch.zeros_like(vars2d)
        gppl = dppl_out.contiguous().float() if dppl_out is not None else torch.zeros(1, device=dev)
        dlogits = K.vq_bwd(logits, gumbel, tau, codes, vars2d, dqc, eproj, dppl, gppl, dvars, N, G, V, D, training).view(N, G * V)
        if tv is not None:
            _ready(pvars)
        dbp = acc_colsum(pbp, dlogits)
        dl = dlogits if dt == torch.float32 else K.cast(dlogits, dt)
        dwp = acc_wgrad(pwp, dl, y, G * V, Cin, N)
        dy = _dgrad(dl, wpc, ctx.wts[0], N, Cin, G * V)
        return dy, dwp, dbp, (None if tv is not None else dvars.view(vshape)), None, None, None, None, None


# ------------------------------------------------------------------ contrastive loss
class ContrastiveXEFn(torch.autograd.Function):
    """CE([cos(c, q+), cos(c, q-_1..N)], 0) over the masked frames (audio8/wav2vec2.py:384-390): the mean (default) or,
    with mean=False, the f32 [1] sum that PretrainLossFn divides itself."""

    @staticmethod
    def forward(ctx, ctx_rows, lat_rows, neg_idx, rows_per_group=0, mean=True):
        logits, loss_sum, inv_norms = K.contrastive_fwd(ctx_rows.contiguous(), lat_rows.contiguous(), neg_idx)
        ctx.save_for_backward(ctx_rows, lat_rows, neg_idx, logits, inv_norms)
        ctx.rows_per_group, ctx.mean = rows_per_group, mean
        rows = ctx_rows.shape[0]
        ctx.mark_non_differentiable(logits)
        ctx.set_materialize_grads(False)         # (no zero-filled [rows, N+1] "gradient" of the logits handed to backward)
        return (loss_sum[0] / rows if mean else loss_sum), logits

    @staticmethod
    def backward(ctx, dloss, _dlogits):
        c, q, neg_idx, logits, inv_norms = ctx.saved_tensors
        rows = c.shape[0]
        if dloss is None:
            return None, None, None, None, None
        gscale = (dloss.float() / rows).reshape(1).contiguous() if ctx.mean else dloss.reshape(1)
        dctx, dlat = K.contrastive_bwd(c.contiguous(), q.contiguous(), neg_idx, logits, gscale, ctx.rows_per_group, inv_norms)
        if c.dtype != torch.float32:
            dctx, dlat = K.cast(dctx, c.dtype), K.cast(dlat, q.dtype)
        return dctx, dlat, None, None, None


class PretrainLossFn(torch.autograd.Function):
    """w_xe * (xe_sum / rows) + w_div * (n_vars - prob_ppl) / n_vars as ONE launch each way (audio8/wav2vec2.py:390-392);
    xe_sum, prob_ppl: f32 one-element device tensors."""

    @staticmethod
    def forward(ctx, xe_sum, ppl, rows, n_vars, w_xe, w_div):
        ctx.cfg = (int(rows), float(n_vars), float(w_xe), float(w_div))
        return K.loss_combine(xe_sum, ppl, *ctx.cfg)

    @staticmethod
    def backward(ctx, dloss):
        dxe, dppl = K.loss_combine_bwd(dloss.contiguous().float(), *ctx.cfg)
        return dxe, dppl, None, None, None, None


# ------------------------------------------------------------------ log-softmax head + CTC
class LogSoftmaxFn(torch.autograd.Function):
    """f32 log_softmax over the first `cols` entries of the last axis (the remainder is the head GEMM's padding)."""

    @staticmethod
    def forward(ctx, x, cols=None):
        shp = x.shape
        cols = shp[-1] if cols is None else cols
        y = K.log_softmax_fwd(x.reshape(-1, shp[-1]).contiguous(), cols)
        ctx.save_for_backward(y)
        ctx.dt, ctx.shp = x.dtype, shp
        return y.view(*shp[:-1], cols)

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return K.log_softmax_bwd(y, dy.reshape(y.shape).contiguous().float(), ctx.dt, ctx.shp[-1]).view(ctx.shp), None


class CTCFn(torch.autograd.Function):
    """Per-utterance CTC negative log-likelihoods (zero_infinity applied) from log-probs [T,B,V]."""

    @staticmethod
    def forward(ctx, lp, flat_i32, off_i32, in_len_i32, Smax, blank, zero_infinity):
        nll, grad = K.ctc_fwd_bwd(lp.contiguous().float(), flat_i32, off_i32, in_len_i32, Smax, blank, zero_infinity)
        ctx.save_for_backward(grad)
        return nll

    @staticmethod
    def backward(ctx, dnll):
        (grad,) = ctx.saved_tensors
        return grad * dnll.view(1, -1, 1), None, None, None, None, None, None


def host_i32(a, device):
    return host_to_device(a, np.int32)
