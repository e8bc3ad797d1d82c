"""MI355X-native mirror of audio8/wav2vec2.py: same factories, class names, constructor arguments,
forward contracts and state_dict keys, so audio8's pretrain.py / train.py drive it unchanged
(SURVEY 8b).  All tensor arithmetic runs in liba8hip.so through audio8_amd.ops; host-side index
sampling (span masks, negatives) stays in numpy with the reference's exact RNG call order.

Deliberate, documented differences from the reference implementation (not from its results):
  * activations are channels-last [B, T, C] end to end (the reference transposes in and out of
    [B, C, T] around its Conv1d stack); the module API never exposed the conv layout;
  * parity knobs for the absent eight_mile source live in audio8_amd.layers.
"""
import contextlib
import math
import collections
import os
from collections import namedtuple
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .layers import Conv1DSame, Dense, TransformerEncoderStack, pytorch_conv1d, pytorch_linear

# audio8/wav2vec2.py:26-35
CONV_FEATURES = {
    16: [(512, 10, 5), (512, 3, 2), (512, 3, 2), (512, 3, 2), (512, 3, 2), (512, 2, 2), (512, 2, 2)],
    8: [(512, 10, 5), (512, 3, 2), (512, 3, 2), (512, 3, 2), (512, 2, 2), (512, 2, 2)],
}
START_TEMP = 2
END_TEMP = 0.5
TEMP_DECAY_FACTOR = 0.999995
XE_WGT = 0.1
DIVERSITY_WGT = 10

CheckpointMapping = namedtuple('CheckpointMapping', ['nested', 'flat'])
# optional: sample the negatives right after the encoder layer-drop draws (same RNG order, earlier on the host)


def _nested_map(src_prefix: str, dst_prefix: str) -> Dict[str, str]:
    """fairseq encoder-layer names -> audio8/eight_mile names (audio8/wav2vec2.py:39-57,109-127).
    fairseq's self_attn_layer_norm is the post-attention LN (ln2), final_layer_norm the post-FFN LN (ln1)."""
    pairs = [("self_attn.k_proj", "self_attn.w_K.layer"), ("self_attn.v_proj", "self_attn.w_V.layer"),
             ("self_attn.q_proj", "self_attn.w_Q.layer"), ("self_attn.out_proj", "self_attn.w_O.layer"),
             ("self_attn_layer_norm", "ln2"), ("fc1", "ffn.0.layer"), ("fc2", "ffn.3.layer"), ("final_layer_norm", "ln1")]
    m = {}
    for s, d in pairs:
        for leaf in ("weight", "bias"):
            m[f"{src_prefix}encoder.layers.{{}}.{s}.{leaf}"] = f"{dst_prefix}encoder.transformer.encoders.{{}}.{d}.{leaf}"
    return m


def _ctc_flat_map(n_conv: int) -> Dict[str, str]:
    """audio8/wav2vec2.py:60-104 (16 kHz: 7 conv layers, 8 kHz: 6)."""
    s, d = "w2v_encoder.w2v_model.", "encoder."
    m = {s + "post_extract_proj.weight": d + "proj_to_input.layer.weight", s + "post_extract_proj.bias": d + "proj_to_input.layer.bias",
         s + "encoder.layer_norm.weight": d + "encoder.ln.weight", s + "encoder.layer_norm.bias": d + "encoder.ln.bias"}
    for leaf in ("bias", "weight_g", "weight_v"):
        m[s + f"encoder.pos_conv.0.{leaf}"] = d + f"encoder.pos_conv.conv.1.{leaf}"
    m[s + "feature_extractor.conv_layers.0.0.weight"] = d + "feature_extractor.conv_layers.0.0.weight"
    m[s + "feature_extractor.conv_layers.0.2.weight"] = d + "feature_extractor.conv_layers.0.2.weight"
    m[s + "feature_extractor.conv_layers.0.2.bias"] = d + "feature_extractor.conv_layers.0.2.bias"
    for i in range(1, n_conv):
        m[s + f"feature_extractor.conv_layers.{i}.0.weight"] = d + f"feature_extractor.conv_layers.{i}.0.weight"
    m[s + "mask_emb"] = d + "mask_emb"
    m[s + "layer_norm.weight"] = d + "layer_norm.weight"
    m[s + "layer_norm.bias"] = d + "layer_norm.bias"
    m["w2v_encoder.proj.weight"] = "proj.weight"
    m["w2v_encoder.proj.bias"] = "proj.bias"
    return m


W2V_CTC_NESTED_MAP = _nested_map("w2v_encoder.w2v_model.", "encoder.")
W2V2_CTC_FLAT_MAP_16 = _ctc_flat_map(7)
W2V2_CTC_FLAT_MAP_8 = _ctc_flat_map(6)
W2V_NESTED_MAP = _nested_map("", "")
W2V2_FLAT_MAP = {          # audio8/wav2vec2.py:131-145
    'post_extract_proj.weight': 'proj_to_input.layer.weight', 'post_extract_proj.bias': 'proj_to_input.layer.bias',
    'project_q.weight': 'project_q.layer.weight', 'project_q.bias': 'project_q.layer.bias',
    'final_proj.weight': 'final_proj.layer.weight', 'final_proj.bias': 'final_proj.layer.bias',
    'encoder.layer_norm.weight': 'encoder.ln.weight', 'encoder.layer_norm.bias': 'encoder.ln.bias',
    'encoder.pos_conv.0.bias': 'encoder.pos_conv.conv.1.bias', 'encoder.pos_conv.0.weight_g': 'encoder.pos_conv.conv.1.weight_g',
    'encoder.pos_conv.0.weight_v': 'encoder.pos_conv.conv.1.weight_v',
}
W2V_MAP = CheckpointMapping(nested=W2V_NESTED_MAP, flat=W2V2_FLAT_MAP)
W2V_CTC_MAP_16 = CheckpointMapping(nested=W2V_CTC_NESTED_MAP, flat=W2V2_CTC_FLAT_MAP_16)
W2V_CTC_MAP_8 = CheckpointMapping(nested=W2V_CTC_NESTED_MAP, flat=W2V2_CTC_FLAT_MAP_8)
W2V_CTC_MAP = {8: W2V_CTC_MAP_8, 16: W2V_CTC_MAP_16}


def convert_keys(num_layers: int, d: Dict, nested_layer_map: Dict = W2V_NESTED_MAP, flat_map: Dict = W2V2_FLAT_MAP) -> Dict:
    """audio8/wav2vec2.py:154-168: rename fairseq keys; every mapped key must be present (KeyError otherwise)."""
    m = {}
    for i in range(num_layers):
        for k, v in nested_layer_map.items():
            m[v.format(i)] = d.pop(k.format(i))
    for k, v in flat_map.items():
        m[v] = d.pop(k)
    m.update(d)
    return m


def load_fairseq_bin(w2v: nn.Module, bin_file: str, ctc: bool = False, sr: int = 16):
    """audio8/wav2vec2.py:171-186."""
    if ctc:
        checkpoint_mapping = W2V_CTC_MAP[sr]
        transformer = w2v.encoder.encoder.transformer
    else:
        checkpoint_mapping = W2V_MAP
        transformer = w2v.encoder.transformer
    d = torch.load(bin_file)["model"]
    mapped_keys = convert_keys(len(transformer.encoders), d, checkpoint_mapping.nested, checkpoint_mapping.flat)
    unknown_keys = w2v.load_state_dict(mapped_keys, strict=False)
    return {'missing': list(unknown_keys.missing_keys), 'unexpected': unknown_keys.unexpected_keys}


def create_mask(shape: Tuple[int, int], p_start: float = 0.65, mask_length: int = 10) -> np.ndarray:
    """Host span-mask sampler with the reference's global-numpy-RNG call order (audio8/wav2vec2.py:189-216):
    rand() once; per row choice(T - min_len, num_mask, replace=False); per over-long row a second choice()."""
    bsz, T = shape
    mask = np.full((bsz, T), False)
    num_mask = int(p_start * T / float(mask_length) + np.random.rand())
    if num_mask == 0:
        return mask
    rows = []
    span = np.arange(mask_length)
    for _ in range(bsz):
        min_len = mask_length
        if T - min_len <= num_mask:
            min_len = T - num_mask - 1
        starts = np.random.choice(T - min_len, num_mask, replace=False)
        idc = (starts[:, None] + span[None, :]).reshape(-1)
        rows.append(np.unique(idc[idc < T]))
    keep = min(len(r) for r in rows)
    for i, r in enumerate(rows):
        if len(r) > keep:
            r = np.random.choice(r, keep, replace=False)
        mask[i, r] = True
    return mask


def create_model(sample_rate=16, num_vq_vars=320, num_vq_groups=2, d_model=768, num_heads=12, num_layers=12, dropout=0.1,
                 d_ff=None, final_dim=256, dropout_input=0.1, dropout_features=0.1, timestep_masking=0.65, channel_masking=0.0,
                 timestep_mask_len=10, channel_mask_len=64, layer_drop=0.0, **kwargs):
    """audio8/wav2vec2.py:219-259."""
    return Wav2Vec2Model(CONV_FEATURES[sample_rate], num_vq_vars, START_TEMP, END_TEMP, TEMP_DECAY_FACTOR, num_vq_groups, d_model,
                         num_heads, num_layers, dropout, d_ff, final_dim, dropout_input, dropout_features, timestep_masking,
                         channel_masking, timestep_mask_len, channel_mask_len, layer_drop)


def create_acoustic_model(num_labels, sample_rate=16, d_model=768, num_heads=12, num_layers=12, dropout=0.1, d_ff=None,
                          dropout_input=0.0, timestep_masking=0.5, channel_masking=0.1, timestep_mask_len=10, channel_mask_len=64,
                          layer_drop=0.0, freeze_fx=True, **kwargs):
    """audio8/wav2vec2.py:262-296."""
    return Wav2Vec2AcousticModel(num_labels, CONV_FEATURES[sample_rate], d_model, num_heads, num_layers, dropout, d_ff, dropout_input,
                                 0.0, timestep_masking, channel_masking, timestep_mask_len, channel_mask_len, layer_drop, freeze_fx)


class Sampler:
    """Negative index sampler (audio8/wav2vec2.py:955-976): same numpy draw, same `>= own -> +1` shift."""

    def __init__(self, n_negatives=100):
        self.n_negatives = n_negatives

    def indices(self, B: int, T: int) -> np.ndarray:
        """int32 [B, n_negatives * T] flat row indices.  Drawn by the library's host routine on numpy's own MT19937 state
        (a8_host_negative_indices: same values, same state afterwards as the numpy statement in indices_numpy, 5x faster -- this draw
        is a fifth of the step's host time at cfg2); T < 3 or a foreign bit generator: the numpy statement itself."""
        state = np.random.get_state()
        if T < 3 or state[0] != "MT19937" or B * T >= 2 ** 31:
            return self.indices_numpy(B, T)
        import ctypes as C
        from . import _lib as L
        key = np.ascontiguousarray(state[1], dtype=np.uint32).copy()
        pos = C.c_int(int(state[2]))
        out = np.empty((B, self.n_negatives * T), dtype=np.int32)
        L.check(L.lib().a8_host_negative_indices(key.ctypes.data, C.addressof(pos), B, T, self.n_negatives, out.ctypes.data),
                "a8_host_negative_indices")
        np.random.set_state((state[0], key, pos.value, state[3], state[4]))
        return out

    def indices_numpy(self, B: int, T: int) -> np.ndarray:
        """The reference's statement (audio8/wav2vec2.py:962-970)."""
        own = np.repeat(np.arange(T), self.n_negatives)
        idx = np.random.randint(0, T - 1, (B, self.n_negatives * T))
        idx += idx >= own[None, :]
        return idx + (np.arange(B) * T)[:, None]

    def negatives(self, y):
        """Reference-shaped API: (negs [N,B,T,C], idx [B, N*T]); the loss itself never materialises negs."""
        B, T, C = y.shape
        idx = torch.from_numpy(self.indices(B, T).astype(np.int64))
        negs = y.reshape(-1, C)[idx.view(-1).to(y.device)].view(B, T, self.n_negatives, C).permute(2, 0, 1, 3)
        return negs, idx


class Wav2Vec2Loss(nn.Module):
    """audio8/wav2vec2.py:371-392: 0.1 * CE(cosine logits) + 10 * (n_vars - prob_ppl) / n_vars."""

    def __init__(self, n_vars, n_negatives=100):
        super().__init__()
        self.n_vars = n_vars
        self.sample = Sampler(n_negatives)
        self.last_logits = None

    def __call__(self, model, features):
        outputs, latents, gs_probs, time_mask = model(features)
        B, T, C = outputs.shape
        Tm = latents.shape[1]
        rows = _mask_row_index(time_mask)
        ctx = ops.GatherRowsFn.apply(outputs.reshape(B * T, C), rows)
        plan = getattr(time_mask, "_a8_plan", None)
        if plan is not None and plan.neg_idx is not None and plan.neg_shape == (B * Tm, self.sample.n_negatives):
            neg_idx = plan.neg_idx                       # drawn ahead of time, in the reference's RNG order
        else:
            if plan is not None and plan.neg_idx is not None:
                raise RuntimeError("a8hip: prefetched negatives do not match this loss (shape %s vs %s)"
                                   % (plan.neg_shape, (B * Tm, self.sample.n_negatives)))
            neg_idx = ops.host_i32(self.sample.indices(B, Tm).reshape(B * Tm, self.sample.n_negatives), outputs.device)
        xe_sum, logits = ops.ContrastiveXEFn.apply(ctx, latents.reshape(B * Tm, C), neg_idx, Tm, False)   # negatives stay inside the utterance
        self.last_logits = logits
        return ops.PretrainLossFn.apply(xe_sum, gs_probs.reshape(1), B * Tm, self.n_vars, XE_WGT, DIVERSITY_WGT).reshape(())


def create_loss(n_vars, n_negatives):
    return Wav2Vec2Loss(n_vars, n_negatives)


def _mask_row_index(time_mask: torch.Tensor) -> torch.Tensor:
    """int32 row indices (b*T + t) of the True entries of a bool [B,T] mask, in row-major order."""
    cached = getattr(time_mask, "_a8_rows", None)
    if cached is not None:
        return cached
    return torch.nonzero(time_mask.reshape(-1)).flatten().to(torch.int32)


def _mask_to_device(mask_np: np.ndarray, device):
    """numpy bool [B,T] -> (bool tensor on device, uint8 flat view, int32 row indices); one H2D each."""
    m = ops.host_to_device(mask_np, np.bool_)
    rows = ops.host_i32(np.flatnonzero(mask_np.reshape(-1)), device)
    m._a8_rows = rows
    return m, m.reshape(-1).view(torch.uint8), rows


class HostDraws:
    """Every global-numpy-RNG draw of one pretrain step, already on the device: span mask, per-layer layer-drop draws,
    negative indices -- taken in the reference's order (audio8/wav2vec2.py:937 -> eight_mile encoder stack -> :384)."""

    def __init__(self, key, mask_dev, layer_draws, neg_idx, neg_shape):
        self.key, self.mask_dev, self.layer_draws, self.neg_idx, self.neg_shape = key, mask_dev, layer_draws, neg_idx, neg_shape


_PLANS = collections.deque()


def conv_out_frames(conv_specs, n_samples: int) -> int:
    for _, k, s in conv_specs:
        n_samples = (n_samples - k) // s + 1
    return n_samples


def prefetch_host_draws(model, loss_fn, batch_shape, device="cuda"):
    """Take the NEXT training step's host-RNG draws now and upload them asynchronously.

    The reference draws mask / layer-drop / negatives inline during forward and loss (2-3 ms of numpy per step at
    8 x 250k samples, with the GPU idle behind it).  None of them depends on device results, only on shapes, so a
    training loop can call this right after `loss.backward()` -- while the host would otherwise just wait in
    `loss.item()` -- and the next `loss_fn(model, x)` consumes the plan instead of drawing.  `batch_shape` is the shape of
    the NEXT batch (the one the next forward will see), not of the one just processed.  The order of calls on the
    global numpy RNG is exactly the reference's (all of step n, then all of step n+1), so sampled indices stay
    bit-identical to a run without prefetching.  Training mode only; if the next forward turns out to have another shape
    the draws are undone (RNG state restored) and that forward draws inline."""
    if not isinstance(model, Wav2Vec2Model) or not model.training:
        return None
    B, L = int(batch_shape[0]), int(batch_shape[-1])
    T = conv_out_frames(model.feature_extractor.conv_specs, L)
    rng_before = np.random.get_state()                  # lets a forward of another shape undo these draws (see _take_plan)
    mask_np = create_mask((B, T), p_start=model.timestep_masking, mask_length=model.timestep_mask_len)
    stack = model.encoder.transformer
    draws = [np.random.random() for _ in stack.encoders] if stack.layerdrop_rng_draws else None
    Tm = int(mask_np[0].sum())
    n_neg = loss_fn.sample.n_negatives
    idx = loss_fn.sample.indices(B, Tm).reshape(B * Tm, n_neg)
    # uploads go on their own stream: on the compute stream they would queue behind the whole backward and sit between
    # it and the loss read-back / optimizer
    global _UPLOAD_STREAM
    if _UPLOAD_STREAM is None:
        _UPLOAD_STREAM = torch.cuda.Stream()
    with torch.cuda.stream(_UPLOAD_STREAM):
        plan = HostDraws((id(model), B, T), _mask_to_device(mask_np, device), draws, ops.host_i32(idx, device), (B * Tm, n_neg))
        plan.ready = torch.cuda.Event()
        plan.ready.record(_UPLOAD_STREAM)
    plan.rng_before = rng_before
    _PLANS.append(plan)
    return plan


_UPLOAD_STREAM = None


def _take_plan(model, B, T):
    if not _PLANS:
        return None
    plan = _PLANS.popleft()
    if plan.key != (id(model), B, T):                   # (train / eval does not matter: the forward draws the same things in both)
        # The plan was drawn for another batch shape (the reference loader's crop-to-min batches change [B, L] almost every
        # step, audio8/data.py:409-426; or an eval pass on another shape came in between).  Rewind the global numpy RNG to
        # where it stood before the plan's draws and let this forward draw inline: indices stay bit-identical to a run
        # that never prefetched.
        np.random.set_state(plan.rng_before)
        _PLANS.clear()
        return None
    cur = torch.cuda.current_stream()
    cur.wait_event(plan.ready)
    for t in (plan.mask_dev[0], plan.mask_dev[2], plan.neg_idx):     # allocated on the upload stream, consumed here
        t.record_stream(cur)
    return plan


class ConvFeatureExtractionModel(nn.Module):
    """audio8/wav2vec2.py:399-456; parameters in the reference's Sequential slots (`conv_layers.{i}.0`, `conv_layers.0.2`)."""

    def __init__(self, conv_layers, dropout: float = 0.0, conv_bias: bool = False):
        super().__init__()
        if conv_bias or dropout != 0.0:
            raise NotImplementedError("a8hip conv extractor: conv_bias=False, dropout=0 (what audio8 instantiates)")
        self.conv_specs = [tuple(c) for c in conv_layers]
        if self.conv_specs[0] != (512, 10, 5):
            raise NotImplementedError("a8hip conv layer 0 is built for Conv1d(1, 512, k=10, s=5)")
        self.conv_layers = nn.ModuleList()
        in_d = 1
        for i, (dim, k, stride) in enumerate(self.conv_specs):
            mods = [pytorch_conv1d(in_d, dim, k, initializer="kaiming", stride=stride, bias=False), nn.Dropout(p=dropout)]
            if i == 0:
                mods.append(nn.GroupNorm(dim, dim, affine=True))
            mods.append(nn.GELU())
            self.conv_layers.append(nn.Sequential(*mods))
            in_d = dim

    def forward(self, x):
        """x [B, L] -> channels-last features [B, T, 512]."""
        gn = self.conv_layers[0][2]
        ws = [blk[0].weight for blk in self.conv_layers]
        return ops.ConvStackFn.apply(x, gn.weight, gn.bias, self.conv_specs, *ws)


class GumbelVectorQuantizer(nn.Module):
    """audio8/wav2vec2.py:459-576, including the (min,max) temperature slot mix-up that pins tau = 0.5."""

    def __init__(self, dim, num_vars, min_temperature, max_temperature, temperature_decay, num_groups, vq_dim):
        super().__init__()
        self.num_groups = num_groups
        self.input_dim = dim
        self.num_vars = num_vars
        assert vq_dim % num_groups == 0, f"dim {vq_dim} must be divisible by groups {num_groups} for concatenation"
        var_dim = vq_dim // num_groups
        self.vars = nn.Parameter(torch.FloatTensor(1, num_groups * num_vars, var_dim))
        nn.init.uniform_(self.vars)
        self.weight_proj = nn.Linear(self.input_dim, num_groups * num_vars)
        nn.init.normal_(self.weight_proj.weight, mean=0, std=1)
        nn.init.zeros_(self.weight_proj.bias)
        self.max_temperature = max_temperature
        self.min_temperature = min_temperature
        self.temperature_decay = temperature_decay
        self.curr_temperature = self.max_temperature
        self.gumbel_override = None       # tests inject the Gumbel noise here ([B*Tm*G, V])
        self.last_codes = None

    def set_num_updates(self, num_updates):
        self.curr_temperature = max(self.max_temperature * self.temperature_decay ** num_updates, self.min_temperature)

    def forward(self, x):
        bsz, tsz, fsz = x.shape
        q, ppl, codes = ops.VQFn.apply(x.reshape(bsz * tsz, fsz), self.weight_proj.weight, self.weight_proj.bias, self.vars,
                                       self.num_groups, self.num_vars, float(self.curr_temperature), self.training, self.gumbel_override)
        self.last_codes = codes
        return q.view(bsz, tsz, -1), ppl.reshape(())


class AudioTransformerEncoder(nn.Module):
    """audio8/wav2vec2.py:579-646."""

    def __init__(self, num_heads: int, d_model: int, pdrop: float, layers: int = 1, activation: str = "gelu", d_ff: Optional[int] = None,
                 conv_pos_kernel: int = 128, conv_groups: int = 16, layer_drop: float = 0.0, **kwargs):
        super().__init__()
        self.d_model = d_model
        self.conv_pos_kernel = conv_pos_kernel
        self.conv_groups = conv_groups
        self.pdrop = pdrop
        std = math.sqrt((4 * (1.0 - pdrop)) / (self.conv_pos_kernel * self.d_model))
        self.pos_conv = Conv1DSame(d_model, d_model, self.conv_pos_kernel, activation="gelu", groups=self.conv_groups, unif=std,
                                   initializer="normal").weight_norm_()
        if not d_ff:
            d_ff = 4 * d_model
        self.transformer = TransformerEncoderStack(num_heads=num_heads, d_model=d_model, pdrop=pdrop, layers=layers, activation=activation,
                                                   layer_norms_after=True, d_ff=d_ff, layer_drop=layer_drop)
        self.ln = nn.LayerNorm(self.d_model)

    def forward(self, x, pad_mask=None):
        return self.extract_features(x, pad_mask)

    def extract_features(self, x, pad_mask=None):
        B, T, d = x.shape
        if pad_mask is not None:
            x = ops.ZeroFn.apply(x, pad_mask.reshape(-1).to(torch.uint8).contiguous(), None)      # x[~pad_mask] = 0
        x = self.pos_conv.forward_residual(x)                                                     # x += gelu(pos_conv(x))
        x = ops.layer_norm(x.reshape(B * T, d), self.ln.weight, self.ln.bias, self.ln.eps).view(B, T, d)
        x = ops.dropout(x, self.pdrop, self.training)
        mask = pad_mask.unsqueeze(1).unsqueeze(1) if pad_mask is not None else None
        return self.transformer((x, mask))


class Wav2Vec2Encoder(nn.Module):
    """Fine-tuning encoder (audio8/wav2vec2.py:649-723)."""

    def __init__(self, conv_features=CONV_FEATURES[16], d_model=768, num_heads=12, num_layers=12, dropout=0.1, d_ff=None,
                 dropout_input=0.1, dropout_features=0.0, timestep_masking=0.5, channel_masking=0.1, timestep_mask_len=10,
                 channel_mask_len=64, layer_drop=0.0, freeze_fx=True):
        super().__init__()
        fx_dsz = conv_features[-1][0]
        self.layer_norm = torch.nn.LayerNorm(fx_dsz)
        self.p_dropout_input = dropout_input
        self.p_dropout_features = dropout_features
        self.feature_extractor = ConvFeatureExtractionModel(conv_features)
        self.proj_to_input = Dense(fx_dsz, d_model)
        self.encoder = AudioTransformerEncoder(num_heads, d_model, dropout, num_layers, d_ff=d_ff, layer_drop=layer_drop)
        self.mask_emb = nn.Parameter(torch.FloatTensor(d_model).uniform_())
        self.timestep_masking = timestep_masking
        self.channel_masking = channel_masking
        self.timestep_mask_len = timestep_mask_len
        self.channel_mask_len = channel_mask_len
        self.output_dim = d_model
        self.freeze_fx = freeze_fx

    def forward(self, x, pad_mask=None):
        with torch.no_grad() if self.freeze_fx else contextlib.ExitStack():
            fx = self.feature_extractor(x)                                                        # [B, T, 512]
        B, T, C0 = fx.shape
        features = ops.layer_norm(fx.reshape(B * T, C0), self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps)
        if pad_mask is not None:
            extra = pad_mask.size(1) % T
            if extra > 0:
                pad_mask = pad_mask[:, :-extra]
            pad_mask = pad_mask.view(pad_mask.size(0), T, -1).all(-1)
        features = self.proj_to_input(features)                                                   # [B*T, d]
        C = features.shape[-1]
        features = ops.dropout(features, self.p_dropout_input, self.training)
        if self.training and self.timestep_masking > 0.0:
            time_mask = create_mask((B, T), p_start=self.timestep_masking, mask_length=self.timestep_mask_len)
            _, tm_u8, _ = _mask_to_device(time_mask, x.device)
            features = ops.MaskFillRowsFn.apply(features, tm_u8, self.mask_emb)
        features = features.view(B, T, C)
        if self.training and self.channel_masking > 0.0:
            channel_mask = create_mask((B, C), p_start=self.channel_masking, mask_length=self.channel_mask_len)
            cm = torch.from_numpy(channel_mask).to(x.device).view(torch.uint8)
            features = ops.ZeroFn.apply(features, None, cm)
        x = self.encoder(features, pad_mask)
        return x, pad_mask


class Wav2Vec2AcousticModel(nn.Module):
    """audio8/wav2vec2.py:726-770: encoder -> Linear(d, num_labels) -> log_softmax."""

    def __init__(self, num_labels, conv_features=CONV_FEATURES[16], d_model=768, num_heads=12, num_layers=12, dropout=0.1, d_ff=None,
                 dropout_input=0.0, dropout_features=0.0, timestep_masking=0.5, channel_masking=0.1, timestep_mask_len=10,
                 channel_mask_len=64, layer_drop=0.0, freeze_fx=True):
        super().__init__()
        self.encoder = Wav2Vec2Encoder(conv_features, d_model, num_heads, num_layers, dropout, d_ff, dropout_input, dropout_features,
                                       timestep_masking, channel_masking, timestep_mask_len, channel_mask_len, layer_drop, freeze_fx=freeze_fx)
        self.proj = pytorch_linear(d_model, num_labels)
        self.freeze = True

    def forward(self, x, pad_mask=None):
        with torch.no_grad() if self.freeze else contextlib.ExitStack():
            encoded, pad_mask = self.encoder(x, pad_mask)
        w, b = self.proj.weight, self.proj.bias
        V = w.shape[0]
        if V % 8:                                   # any vocabulary size (letters or BPE): the GEMM runs on a label axis padded
            pad = 8 - V % 8                         # to a multiple of 8 with zero rows; log_softmax reads the first V columns only
            w = torch.cat([w, w.new_zeros(pad, w.shape[1])], 0)
            b = torch.cat([b, b.new_zeros(pad)], 0)
        logits = ops.linear(encoded, w, b)
        return ops.LogSoftmaxFn.apply(logits, V), pad_mask


class Wav2Vec2Model(nn.Module):
    """Pretraining model (audio8/wav2vec2.py:871-952)."""

    def __init__(self, conv_features=CONV_FEATURES[16], num_vq_vars=320, start_temp=START_TEMP, end_temp=END_TEMP,
                 temp_decay_factor=TEMP_DECAY_FACTOR, num_vq_groups=2, d_model=768, num_heads=12, num_layers=12, dropout=0.1, d_ff=None,
                 final_dim=256, dropout_input=0.1, dropout_features=0.1, timestep_masking=0.65, channel_masking=0.0,
                 timestep_mask_len=10, channel_mask_len=64, layer_drop=0.0):
        super().__init__()
        fx_dsz = conv_features[-1][0]
        self.layer_norm = torch.nn.LayerNorm(fx_dsz)
        self.p_dropout_input = dropout_input
        self.p_dropout_features = dropout_features
        self.feature_extractor = ConvFeatureExtractionModel(conv_features)
        self.proj_to_input = Dense(fx_dsz, d_model)
        self.quantizer = GumbelVectorQuantizer(fx_dsz, num_vq_vars, start_temp, end_temp, temp_decay_factor, num_vq_groups, final_dim)
        self.encoder = AudioTransformerEncoder(num_heads, d_model, dropout, num_layers, d_ff=d_ff, layer_drop=layer_drop)
        self.project_q = Dense(final_dim, final_dim)
        self.final_proj = Dense(d_model, final_dim)
        self.timestep_masking = timestep_masking
        self.channel_masking = channel_masking
        self.timestep_mask_len = timestep_mask_len
        self.channel_mask_len = channel_mask_len
        self.mask_emb = nn.Parameter(torch.FloatTensor(d_model).uniform_())
        if channel_masking > 0.0:
            # the reference's channel-mask branch is a latent defect (invalid view, audio8/wav2vec2.py:941-944), dead at the default 0.0
            raise NotImplementedError("Wav2Vec2Model: channel_masking > 0 is unreachable in the reference as well")

    def set_num_updates(self, s):
        self.quantizer.set_num_updates(s)

    def forward(self, x):
        fx = self.feature_extractor(x)                                                            # [B, T, 512] channels-last (:928-929)
        B, T, C0 = fx.shape
        unmasked = ops.layer_norm(fx.reshape(B * T, C0), self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps)   # :930-931
        features = self.proj_to_input(unmasked)                                                   # :932
        features = ops.dropout(features, self.p_dropout_input, self.training)                    # :934
        unmasked_d = ops.dropout(unmasked, self.p_dropout_features, self.training)               # :935
        plan = _take_plan(self, B, T)
        if plan is not None:                                                                      # draws taken by prefetch_host_draws
            time_mask, tm_u8, rows = plan.mask_dev
            time_mask._a8_plan = plan
            ops.PLANNED_LAYER_DRAWS = plan.layer_draws
        else:
            time_mask_np = create_mask((B, T), p_start=self.timestep_masking, mask_length=self.timestep_mask_len)   # :937
            time_mask, tm_u8, rows = _mask_to_device(time_mask_np, x.device)                      # :938
        features = ops.MaskFillRowsFn.apply(features, tm_u8, self.mask_emb)                      # :939
        Tm = rows.numel() // B
        y = ops.GatherRowsFn.apply(unmasked_d, rows).view(B, Tm, C0)                              # :946
        ctx = self.encoder(features.view(B, T, -1))                                               # :947
        y, vq_probs = self.quantizer(y)                                                           # :948
        y = self.project_q(y)                                                                     # :950
        ctx = self.final_proj(ctx)                                                                # :951
        return ctx, y, vq_probs, time_mask
