"""The slice of the eight_mile (mead-baseline) layer API that audio8's wav2vec2 path uses
(audio8/wav2vec2.py:9-21), re-implemented over the HIP ops: same class names, constructor
arguments, tuple-input conventions and -- crucially -- the same parameter names, so the
state_dict keys of audio8_amd.wav2vec2 models equal the reference's (fairseq key maps,
audio8/wav2vec2.py:39-151).

eight_mile's source is not part of the reference tree (unversioned third-party dependency), so the
behaviours that cannot be read there are explicit knobs (module attributes):
    TransformerEncoder.ln_eps        = 1e-6   (eight_mile LayerNorm eps; fairseq/HF use 1e-5)
    Conv1DSame pad                   = (k//2 - 1, k//2) for an even kernel
    TransformerEncoderStack.layerdrop_rng_draws = True  (one np.random.random() per layer per forward)
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import ops


class Dense(nn.Module):
    """eight_mile Dense: owns an nn.Linear as `.layer` (keys `<name>.layer.weight/bias`)."""

    def __init__(self, insz, outsz, activation=None, unif=0, initializer=None):
        super().__init__()
        self.layer = pytorch_linear(insz, outsz, unif, initializer)
        self.output_dim = outsz
        if activation not in (None, "gelu"):
            raise ValueError("a8hip Dense supports activation None or 'gelu'")
        self.activation = activation

    def forward(self, x):
        y = ops.linear(x, self.layer.weight, self.layer.bias)
        if self.activation == "gelu":
            raise NotImplementedError("fused Dense+GELU is only used inside EncoderLayerFn")
        return y


def pytorch_linear(in_sz, out_sz, unif=0, initializer=None, bias=True):
    """eight_mile pytorch_linear: nn.Linear with xavier-uniform weights (default) and zero bias."""
    l = nn.Linear(in_sz, out_sz, bias=bias)
    if unif > 0:
        l.weight.data.uniform_(-unif, unif)
    elif initializer == "ortho":
        nn.init.orthogonal_(l.weight)
    elif initializer in ("he", "kaiming"):
        nn.init.kaiming_uniform_(l.weight)
    else:
        nn.init.xavier_uniform_(l.weight)
    if bias:
        l.bias.data.zero_()
    return l


def pytorch_conv1d(in_channels, out_channels, fsz, unif=0, padding=0, initializer=None, stride=1, bias=True, groups=1):
    """eight_mile pytorch_conv1d: parameter container + init (the arithmetic is in ops.ConvStackFn/PosConvFn)."""
    c = nn.Conv1d(in_channels, out_channels, fsz, padding=padding, stride=stride, bias=bias, groups=groups)
    if unif > 0:
        c.weight.data.uniform_(-unif, unif)
    elif initializer == "ortho":
        nn.init.orthogonal_(c.weight)
    elif initializer in ("he", "kaiming"):
        nn.init.kaiming_uniform_(c.weight)
    elif initializer == "normal":
        nn.init.normal_(c.weight, mean=0, std=unif)
    else:
        nn.init.xavier_uniform_(c.weight)
    if bias:
        c.bias.data.zero_()
    return c


class _WeightNormConvParams(nn.Module):
    """Parameter holder with nn.utils.weight_norm(name='weight', dim=2) names: bias, weight_g [1,1,k], weight_v."""

    def __init__(self, conv: nn.Conv1d):
        super().__init__()
        v = conv.weight.data
        self.bias = nn.Parameter(conv.bias.data.clone())
        self.weight_g = nn.Parameter(v.norm(2, dim=(0, 1), keepdim=True).clone())
        self.weight_v = nn.Parameter(v.clone())
        self.groups = conv.groups
        self.kernel_size = conv.kernel_size

    def effective_weight(self):
        v = self.weight_v
        return v * (self.weight_g / v.norm(2, dim=(0, 1), keepdim=True))


class Conv1DSame(nn.Module):
    """eight_mile Conv1DSame: `.conv = Sequential(ConstantPad1d, conv, activation)`; audio8 replaces
    `.conv[1]` by its weight-normed version (audio8/wav2vec2.py:609).  Input/output [B, T, C] here
    (channels-last; the reference transposes around it, audio8/wav2vec2.py:634-635)."""

    def __init__(self, in_channels, out_channels, kernel_size, bias=True, groups=1, unif=0.0, initializer=None, activation=None):
        super().__init__()
        end_pad = kernel_size // 2
        start_pad = end_pad - 1 if kernel_size % 2 == 0 else end_pad
        self.pad = (start_pad, end_pad)
        self.conv = nn.Sequential(
            nn.Identity(),
            pytorch_conv1d(in_channels, out_channels, kernel_size, unif=unif, initializer=initializer, bias=bias, groups=groups),
            nn.Identity(),
        )
        if activation != "gelu":
            raise ValueError("a8hip Conv1DSame implements the GELU-activated positional conv only")

    def weight_norm_(self):
        self.conv[1] = _WeightNormConvParams(self.conv[1])
        return self

    def forward_residual(self, x):
        """x + gelu(conv_same(x)) for x [B, T, C]."""
        m = self.conv[1]
        if not isinstance(m, _WeightNormConvParams):
            raise RuntimeError("Conv1DSame: call weight_norm_() first (the audio path always does)")
        return ops.PosConvFn.apply(x, m.weight_v, m.weight_g, m.bias, m.groups, self.pad[0])


class MultiHeadedAttention(nn.Module):
    """Parameter container with eight_mile's names (w_Q, w_K, w_V, w_O: Dense)."""

    def __init__(self, num_heads, d_model, dropout=0.1, scale=True, d_k=None):
        super().__init__()
        if d_k is not None and d_k != d_model // num_heads:
            raise ValueError("a8hip attention needs d_k = d_model / num_heads")
        if d_model // num_heads != 64:
            raise ValueError("a8hip attention kernels are built for head dim 64 (wav2vec2 base and large)")
        self.h = num_heads
        self.w_Q = Dense(d_model, d_model)
        self.w_K = Dense(d_model, d_model)
        self.w_V = Dense(d_model, d_model)
        self.w_O = Dense(d_model, d_model)


class TransformerEncoder(nn.Module):
    """eight_mile TransformerEncoder (one block).  Post-LN when layer_norms_after (the only mode the audio path uses)."""

    ln_eps = 1e-6

    def __init__(self, num_heads, d_model, pdrop, scale=True, activation_type="gelu", d_ff=None, d_k=None,
                 ffn_pdrop=0.0, layer_norms_after=False, layer_norm_eps=1e-6, **kwargs):
        super().__init__()
        if not layer_norms_after:
            raise NotImplementedError("a8hip implements the post-LN block used by AudioTransformerEncoder")
        if activation_type != "gelu" or ffn_pdrop != 0.0:
            raise NotImplementedError("a8hip encoder block: GELU FFN without FFN-internal dropout")
        self.d_model = d_model
        self.d_ff = d_ff if d_ff is not None else 4 * d_model
        self.pdrop = pdrop
        self.ln_eps = layer_norm_eps
        self.self_attn = MultiHeadedAttention(num_heads, d_model, pdrop, scale=scale, d_k=d_k)
        self.ffn = nn.Sequential(Dense(d_model, self.d_ff), nn.Identity(), nn.Identity(), Dense(self.d_ff, d_model))
        self.ln1 = nn.LayerNorm(d_model, eps=layer_norm_eps)
        self.ln2 = nn.LayerNorm(d_model, eps=layer_norm_eps)

    def forward(self, inputs):
        x, mask_u8, B, T = inputs          # x [B*T, d]; mask_u8 [B, T] uint8 (1 = attend) or None
        a = self.self_attn
        p = self.pdrop if self.training else 0.0
        return ops.EncoderLayerFn.apply(
            x, mask_u8, B, T, a.h, p, self.ln_eps,
            a.w_Q.layer.weight, a.w_Q.layer.bias, a.w_K.layer.weight, a.w_K.layer.bias,
            a.w_V.layer.weight, a.w_V.layer.bias, a.w_O.layer.weight, a.w_O.layer.bias,
            self.ffn[0].layer.weight, self.ffn[0].layer.bias, self.ffn[3].layer.weight, self.ffn[3].layer.bias,
            self.ln1.weight, self.ln1.bias, self.ln2.weight, self.ln2.bias)


class TransformerEncoderStack(nn.Module):
    """eight_mile TransformerEncoderStack: `.encoders` ModuleList; final `.ln` is Identity when
    layer_norms_after.  Takes the (x, mask) tuple; mask is the reference's [B,1,1,T] bool (or None)."""

    layerdrop_rng_draws = True

    def __init__(self, num_heads, d_model, pdrop, scale=True, layers=1, activation="gelu", d_ff=None, d_k=None,
                 layer_norms_after=False, layer_norm_eps=1e-6, layer_drop=0.0, **kwargs):
        super().__init__()
        self.encoders = nn.ModuleList()
        self.ln = nn.Identity() if layer_norms_after else nn.LayerNorm(d_model, eps=layer_norm_eps)
        if not layer_norms_after:
            raise NotImplementedError("a8hip implements the post-LN stack used by AudioTransformerEncoder")
        self.output_dim = d_model
        self.layer_drop = layer_drop
        for _ in range(layers):
            self.encoders.append(TransformerEncoder(num_heads, d_model, pdrop, scale, activation, d_ff, d_k,
                                                    layer_norms_after=layer_norms_after, layer_norm_eps=layer_norm_eps))

    def forward(self, inputs):
        x, mask = inputs
        B, T, d = x.shape
        mask_u8 = None
        if mask is not None:
            mask_u8 = mask.reshape(B, T).to(torch.uint8).contiguous()
        ops.wait_weight_prep()                 # the stack's bf16 weight copies may still be building on the side stream
        h = x.reshape(B * T, d)
        # one np.random.random() per layer per forward (eight_mile's layer-drop draw), taken up front -- or handed over by
        # wav2vec2.prefetch_host_draws, which took them in the same order while the GPU was busy with the previous step
        draws = ops.take_layer_draws(len(self.encoders)) if self.layerdrop_rng_draws else [1.0] * len(self.encoders)
        # two half-batches on two streams through the whole stack (ops.HALF_BATCH_STREAMS): half A's launches stay on the current
        # stream, half B's go to an auxiliary one; the streams meet only here, before the first and after the last layer
        aux = ops.half_stream(x.device) if (x.is_cuda and B >= 2 and h.dtype == torch.bfloat16 and h.is_contiguous()) else None
        cur = torch.cuda.current_stream() if aux is not None else None
        if aux is not None:
            aux.wait_stream(cur)
        ops.HALF_STREAM = aux
        edges = aux is not None and ops.HALF_BATCH_BWD and torch.is_grad_enabled() and h.requires_grad
        if edges:
            h = ops.HalfEdgeFn.apply(h, aux, False)          # backward: the caller's stream waits for the second halves
        try:
            for layer, pdrop in zip(self.encoders, draws):
                if self.training and pdrop < self.layer_drop:
                    continue
                h = layer((h, mask_u8, B, T))
        finally:
            ops.HALF_STREAM = None
            if aux is not None:
                cur.wait_stream(aux)
        if edges:
            h = ops.HalfEdgeFn.apply(h, aux, True)           # backward: aux waits for the incoming gradient
        return h.view(B, T, d)


def sequence_mask(lengths, max_len: int = -1):
    """eight_mile.pytorch.layers.sequence_mask: bool [B, max_len], True where index < length."""
    if max_len < 0:
        max_len = int(torch.max(lengths).item())
    row = torch.arange(0, max_len, device=lengths.device).unsqueeze(0)
    return row < lengths.unsqueeze(1)
