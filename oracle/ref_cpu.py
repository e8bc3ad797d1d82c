"""CPU oracle for the wav2vec2 hot path of mead-ml/audio8 -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
module.  The product (audio8_amd/) never does: it fails loudly if its HIP library is missing.

What this is: a plain-PyTorch (fp32, CPU) / numpy restatement of the reference's algorithm
for the path SURVEY.md section 8 scopes, written from the reference's behaviour, each function
citing the reference file:line it follows.  It is functional (weights come in as a dict keyed
by the reference's own state_dict names) so it doubles as a check of checkpoint-key parity.

Pinning status (see DESIGN.md "Oracle"):
  * PINNED by golden vectors produced from the reference's own code (tools/gen_golden.py,
    tests/golden/g1..g6): create_mask, Sampler.negatives, ConvFeatureExtractionModel,
    GumbelVectorQuantizer, Wav2Vec2Loss, CTCLoss.
  * PARITY UNPINNED: everything the reference imports from the third-party, unversioned
    `mead-baseline` package (eight_mile), whose source is not under /root/reference:
    TransformerEncoderStack / MultiHeadedAttention / Dense / Conv1DSame.  Restated from the
    package's public behaviour and the structural constraints of the reference's fairseq key
    maps (audio8/wav2vec2.py:109-145); the unverifiable choices are explicit parameters
    (`ln_eps`, `posconv_pad`, `layerdrop_rng_draws`).
"""
import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

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

# eight_mile defaults recalled in SURVEY 8(a) rows a4/a5 -- unpinned, hence parameters.
DEFAULT_LN_EPS = 1e-6          # eight_mile TransformerEncoder LayerNorm eps
DEFAULT_POSCONV_PAD = (63, 64)  # Conv1DSame: even kernel -> (k//2 - 1, k//2)


# --------------------------------------------------------------------------------------
# a3: span mask sampling (host, numpy global RNG)            audio8/wav2vec2.py:189-216
# --------------------------------------------------------------------------------------
def create_mask(shape: Tuple[int, int], p_start: float = 0.65, mask_length: int = 10) -> np.ndarray:
    """Equal-count span mask.  RNG call order on the global legacy numpy RNG:
    one rand(); per row one choice(T - min_len, num_mask, replace=False); then, per row whose
    unique count exceeds the batch minimum, one choice(row_idx, min_count, replace=False)."""
    bsz, T = shape
    mask = np.zeros((bsz, T), dtype=bool)
    num_mask = int(p_start * T / float(mask_length) + np.random.rand())       # :192
    if num_mask == 0:
        return mask
    rows = []
    for _ in range(bsz):
        min_len = mask_length
        if T - min_len <= num_mask:                                           # :200-201
            min_len = T - num_mask - 1
        starts = np.random.choice(T - min_len, num_mask, replace=False)       # :203
        idc = (starts[:, None] + np.arange(mask_length)[None, :]).reshape(-1)  # :205
        rows.append(np.unique(idc[idc < T]))                                  # :207
    keep = min(len(r) for r in rows)                                          # :209-210
    for i, r in enumerate(rows):
        if len(r) > keep:
            r = np.random.choice(r, keep, replace=False)                      # :213
        mask[i, r] = True
    return mask


# --------------------------------------------------------------------------------------
# a8: negative index sampling (host, numpy global RNG)       audio8/wav2vec2.py:955-976
# --------------------------------------------------------------------------------------
def sample_negative_indices(B: int, Tm: int, n_negatives: int = 100) -> np.ndarray:
    """int64 [B, n_negatives*Tm] row-major over (t, n); values index the flattened [B*Tm]
    latents.  A frame never samples itself; negatives come from the same utterance only."""
    own = np.repeat(np.arange(Tm), n_negatives)                               # :964-966
    idx = np.random.randint(0, Tm - 1, (B, n_negatives * Tm))                 # :967
    idx = idx + (idx >= own[None, :])                                         # :969
    return (idx + (np.arange(B) * Tm)[:, None]).astype(np.int64)              # :970-973


# --------------------------------------------------------------------------------------
# a1: conv feature extractor                                  audio8/wav2vec2.py:399-456
# --------------------------------------------------------------------------------------
def conv_feature_extractor(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str = "feature_extractor.",
                           conv_features=CONV_FEATURES[16], return_all=False):
    """x [B, L] -> [B, 512, T].  Layer 0: conv -> GroupNorm(512 groups) -> exact GELU;
    layers 1..: conv -> GELU.  No conv bias; Dropout(p=0)."""
    h = x.unsqueeze(1)                                                        # :451
    outs = []
    for i, (dim, k, stride) in enumerate(conv_features):
        w = sd[f"{prefix}conv_layers.{i}.0.weight"]
        h = F.conv1d(h, w, None, stride=stride)
        if i == 0:                                                            # :417-423
            h = F.group_norm(h, dim, sd[f"{prefix}conv_layers.0.2.weight"], sd[f"{prefix}conv_layers.0.2.bias"], 1e-5)
        h = F.gelu(h)
        outs.append(h)
    return outs if return_all else h


def conv_out_lengths(L: int, conv_features=CONV_FEATURES[16]):
    out = []
    for (_, k, s) in conv_features:
        L = (L - k) // s + 1
        out.append(L)
    return out


# --------------------------------------------------------------------------------------
# a4: positional conv + encoder wrapper                       audio8/wav2vec2.py:579-646
# --------------------------------------------------------------------------------------
def weight_norm_dim2(g: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """nn.utils.weight_norm(name='weight', dim=2): w = g * v / ||v||, norm over dims (0,1)
    for every kernel tap.  g [1,1,K], v [Cout, Cin/groups, K].       audio8/wav2vec2.py:609"""
    return v * (g / v.norm(2, dim=(0, 1), keepdim=True))


def pos_conv(x: torch.Tensor, sd, prefix="encoder.", groups=16, pad=DEFAULT_POSCONV_PAD):
    """x [B,T,d] -> gelu(conv1d_same(x^T))^T.  eight_mile Conv1DSame pads (k//2-1, k//2) for an
    even kernel (recalled; `pad` is the knob).                          audio8/wav2vec2.py:600-609,634-635"""
    w = weight_norm_dim2(sd[f"{prefix}pos_conv.conv.1.weight_g"], sd[f"{prefix}pos_conv.conv.1.weight_v"])
    h = F.pad(x.transpose(1, 2), pad)
    h = F.conv1d(h, w, sd[f"{prefix}pos_conv.conv.1.bias"], groups=groups)
    return F.gelu(h).transpose(1, 2)


# --------------------------------------------------------------------------------------
# a5: eight_mile TransformerEncoderStack (post-LN)           call site audio8/wav2vec2.py:613-622,644
# --------------------------------------------------------------------------------------
def encoder_layer(x, sd, pfx, num_heads, mask=None, ln_eps=DEFAULT_LN_EPS, pdrop=0.0, training=False,
                  return_attn=False):
    """Post-LN block (`layer_norms_after=True`): h = MHA(x); x = ln2(x + drop(h));
    x = ln1(x + drop(FFN(x))); FFN = Dense(d, d_ff) -> GELU -> Dense(d_ff, d).
    Key names pinned by the fairseq map audio8/wav2vec2.py:109-127 (self_attn_layer_norm -> ln2,
    final_layer_norm -> ln1, fc1 -> ffn.0, fc2 -> ffn.3)."""
    B, T, d = x.shape
    dk = d // num_heads
    lin = lambda t, n: F.linear(t, sd[f"{pfx}{n}.layer.weight"], sd[f"{pfx}{n}.layer.bias"])
    q = lin(x, "self_attn.w_Q").view(B, T, num_heads, dk).transpose(1, 2)
    k = lin(x, "self_attn.w_K").view(B, T, num_heads, dk).transpose(1, 2)
    v = lin(x, "self_attn.w_V").view(B, T, num_heads, dk).transpose(1, 2)
    scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(dk)            # SeqScaledDotProductAttention
    if mask is not None:
        scores = scores.masked_fill(mask == 0, -1e9)
    a = F.softmax(scores, dim=-1)
    a = F.dropout(a, pdrop, training)
    h = torch.matmul(a, v).transpose(1, 2).contiguous().view(B, T, d)
    h = lin(h, "self_attn.w_O")
    x = x + F.dropout(h, pdrop, training)
    x = F.layer_norm(x, (d,), sd[f"{pfx}ln2.weight"], sd[f"{pfx}ln2.bias"], ln_eps)
    h = lin(F.gelu(lin(x, "ffn.0")), "ffn.3")
    x = x + F.dropout(h, pdrop, training)
    x = F.layer_norm(x, (d,), sd[f"{pfx}ln1.weight"], sd[f"{pfx}ln1.bias"], ln_eps)
    return (x, a) if return_attn else x


def audio_transformer_encoder(x, sd, num_layers, num_heads, pad_mask=None, prefix="encoder.",
                              ln_eps=DEFAULT_LN_EPS, posconv_pad=DEFAULT_POSCONV_PAD, pdrop=0.0,
                              training=False, layer_drop=0.0, layerdrop_rng_draws=True):
    """audio8/wav2vec2.py:629-646.  x [B,T,d]; pad_mask bool [B,T] or None."""
    d = x.shape[-1]
    if pad_mask is not None:
        x = x.masked_fill(~pad_mask.unsqueeze(-1), 0.0)                       # :631-632
    x = x + pos_conv(x, sd, prefix, pad=posconv_pad)                          # :634-638
    x = F.layer_norm(x, (d,), sd[f"{prefix}ln.weight"], sd[f"{prefix}ln.bias"], 1e-5)   # :623,639
    x = F.dropout(x, pdrop, training)
    mask = None if pad_mask is None else pad_mask.unsqueeze(1).unsqueeze(1)   # :641-642
    for i in range(num_layers):
        # eight_mile TransformerEncoderStack draws np.random.random() per layer per forward
        # for layer-drop (recalled; advances the numpy RNG that a8 later consumes).
        pdraw = np.random.random() if layerdrop_rng_draws else 1.0
        if training and pdraw < layer_drop:
            continue
        x = encoder_layer(x, sd, f"{prefix}transformer.encoders.{i}.", num_heads, mask, ln_eps, pdrop, training)
    return x


# --------------------------------------------------------------------------------------
# a6: Gumbel vector quantiser                                 audio8/wav2vec2.py:459-576
# --------------------------------------------------------------------------------------
def quantizer_temperature(num_updates: Optional[int] = None) -> float:
    """The reference passes (start=2, end=0.5, decay) into (min, max, decay) slots
    (audio8/wav2vec2.py:460 vs :910-912): tau = 0.5 at construction (:492) and
    set_num_updates(n) gives max(0.5 * decay**n, 2) = 2 (:495-496)."""
    min_t, max_t = START_TEMP, END_TEMP
    if num_updates is None:
        return float(max_t)
    return float(max(max_t * TEMP_DECAY_FACTOR ** num_updates, min_t))


def gumbel_vq(y, sd, prefix="quantizer.", num_groups=2, num_vars=320, tau=0.5, training=True, gumbel=None):
    """y [B,Tm,C] -> (q [B,Tm,vq_dim], prob_ppl scalar).
    Train: hard Gumbel-softmax with straight-through gradient (torch F.gumbel_softmax, hard=True):
    y_soft = softmax((logits + g)/tau); out = onehot(argmax y_soft) - y_soft.detach() + y_soft.
    `gumbel` injects the noise g ([B*Tm*G, V]); None draws it with torch's own recipe.
    Eval: argmax one-hot (:558-562).  prob_ppl pools avg_probs over groups (:554,565)."""
    B, Tm, C = y.shape
    logits = F.linear(y, sd[f"{prefix}weight_proj.weight"], sd[f"{prefix}weight_proj.bias"])   # :551
    logits = logits.view(B * Tm * num_groups, -1)                                                # :553
    avg_probs = torch.softmax(logits.float(), dim=-1).mean(dim=0)                                # :554
    if training:
        if gumbel is None:
            gumbel = -torch.empty_like(logits).exponential_().log()
        y_soft = ((logits + gumbel) / tau).softmax(-1)
        index = y_soft.max(-1, keepdim=True)[1]
        y_hard = torch.zeros_like(logits).scatter_(-1, index, 1.0)
        onehot = y_hard - y_soft.detach() + y_soft
    else:
        index = logits.max(-1, keepdim=True)[1]
        onehot = torch.zeros_like(logits).scatter_(-1, index, 1.0)
    prob_ppl = torch.sum(torch.exp(-torch.sum(avg_probs * torch.log(avg_probs + 1e-7), -1)))     # :565
    cb = sd[f"{prefix}vars"]                                                                     # [1, G*V, D]
    x = onehot.view(B * Tm, num_groups * num_vars, 1) * cb                                       # :569-570
    x = x.view(B * Tm, num_groups, num_vars, -1).sum(-2).view(B, Tm, -1)                         # :572-575
    return x, prob_ppl


# --------------------------------------------------------------------------------------
# a9: contrastive + diversity loss                            audio8/wav2vec2.py:371-392
# --------------------------------------------------------------------------------------
def contrastive_logits(outputs_m, latents, neg_idx):
    """outputs_m [B,Tm,C] (context vectors at masked frames), latents [B,Tm,C], neg_idx int64
    [B, N*Tm] -> logits [Tm*B, N+1] ordered as the reference's transpose(2,0).reshape (:386-387):
    row = t*B + b; column 0 is the positive.  Raw cosine similarity, no temperature, no -inf."""
    B, Tm, C = latents.shape
    N = neg_idx.shape[1] // Tm
    negs = latents.reshape(-1, C)[neg_idx.reshape(-1)].view(B, Tm, N, C).permute(2, 0, 1, 3)     # :974-975
    targets = torch.cat([latents.unsqueeze(0), negs], dim=0)                                      # :384
    logits = torch.cosine_similarity(outputs_m.unsqueeze(0), targets, dim=-1)                     # :385
    return logits.transpose(2, 0).reshape(-1, N + 1)


def wav2vec2_loss(outputs, latents, prob_ppl, time_mask, neg_idx, n_vars):
    """outputs [B,T,C] -> scalar loss = 0.1 * CE(logits, 0) + 10 * (n_vars - ppl)/n_vars."""
    B, T, C = outputs.shape
    om = outputs[time_mask.unsqueeze(-1).expand_as(outputs)].view(B, -1, C)                       # :381
    logits = contrastive_logits(om, latents, neg_idx)
    tgt = torch.zeros(logits.shape[0], dtype=torch.long)
    xe = F.cross_entropy(logits, tgt)                                                             # :390
    return XE_WGT * xe + DIVERSITY_WGT * (n_vars - prob_ppl) / n_vars                             # :389-392


# --------------------------------------------------------------------------------------
# a2/a6/a7: the pretraining model forward                     audio8/wav2vec2.py:927-952
# --------------------------------------------------------------------------------------
def wav2vec2_model_forward(x, sd, *, num_layers=12, num_heads=12, num_vq_groups=2, num_vq_vars=320,
                           timestep_masking=0.65, timestep_mask_len=10, training=True, gumbel=None,
                           time_mask=None, ln_eps=DEFAULT_LN_EPS, posconv_pad=DEFAULT_POSCONV_PAD,
                           layerdrop_rng_draws=True, tau=None, layer_drop=0.0, conv_features=CONV_FEATURES[16]):
    """Dropout-free restatement of Wav2Vec2Model.forward (parity runs use dropout=0).
    layer_drop > 0 (audio8/pretrain.py:47 -> wav2vec2.py:613-622) skips an encoder layer in training whenever its per-layer
    numpy draw falls below it; conv_features = CONV_FEATURES[sample_rate] (:26-29).
    Returns (x [B,T,final], y [B,Tm,final], prob_ppl, time_mask bool [B,T])."""
    fx = conv_feature_extractor(x, sd, conv_features=conv_features).transpose(1, 2)               # :928-929
    feats = F.layer_norm(fx, (fx.shape[-1],), sd["layer_norm.weight"], sd["layer_norm.bias"], 1e-5)   # :930
    unmasked = feats.clone()                                                                      # :931
    feats = F.linear(feats, sd["proj_to_input.layer.weight"], sd["proj_to_input.layer.bias"])     # :932
    B, T, C = unmasked.shape
    if time_mask is None:
        time_mask = create_mask((B, T), timestep_masking, timestep_mask_len)                      # :937
    tm = torch.from_numpy(time_mask)
    feats = torch.where(tm.unsqueeze(-1), sd["mask_emb"].view(1, 1, -1), feats)                   # :939
    y = unmasked[tm].view(B, -1, C)                                                               # :946
    ctx = audio_transformer_encoder(feats, sd, num_layers, num_heads, None, "encoder.", ln_eps, posconv_pad,
                                    training=training, layer_drop=layer_drop,
                                    layerdrop_rng_draws=layerdrop_rng_draws)                      # :947
    q, ppl = gumbel_vq(y, sd, "quantizer.", num_vq_groups, num_vq_vars,
                       quantizer_temperature() if tau is None else tau, training, gumbel)        # :948
    q = F.linear(q, sd["project_q.layer.weight"], sd["project_q.layer.bias"])                     # :950
    ctx = F.linear(ctx, sd["final_proj.layer.weight"], sd["final_proj.layer.bias"])               # :951
    return ctx, q, ppl, time_mask


def pretrain_loss(x, sd, *, n_negatives=100, neg_idx=None, **kw):
    """One pretrain.py loss evaluation (audio8/pretrain.py:178 -> wav2vec2.py:377-392)."""
    ctx, q, ppl, tmask = wav2vec2_model_forward(x, sd, **kw)
    B, Tm, _ = q.shape
    if neg_idx is None:
        neg_idx = sample_negative_indices(B, Tm, n_negatives)
    nv = kw.get("num_vq_groups", 2) * kw.get("num_vq_vars", 320)
    loss = wav2vec2_loss(ctx, q, ppl, torch.from_numpy(tmask), torch.from_numpy(neg_idx), nv)
    return loss, dict(ctx=ctx, q=q, ppl=ppl, time_mask=tmask, neg_idx=neg_idx)


# --------------------------------------------------------------------------------------
# a10: fine-tune encoder + acoustic head                      audio8/wav2vec2.py:696-723,765-770
# --------------------------------------------------------------------------------------
def downsample_pad_mask(pad_mask: torch.Tensor, T: int) -> torch.Tensor:
    """bool [B,L] -> bool [B,T]: drop the L % T tail, view (B,T,-1), all(-1).   :703-708"""
    extra = pad_mask.size(1) % T
    if extra > 0:
        pad_mask = pad_mask[:, :-extra]
    return pad_mask.view(pad_mask.size(0), T, -1).all(-1)


def acoustic_model_forward(x, pad_mask, sd, *, num_layers=12, num_heads=12, time_mask=None, channel_mask=None,
                           ln_eps=DEFAULT_LN_EPS, posconv_pad=DEFAULT_POSCONV_PAD, layerdrop_rng_draws=True,
                           training=False, layer_drop=0.0, conv_features=CONV_FEATURES[16]):
    """Wav2Vec2AcousticModel.forward, dropout-free.  time_mask/channel_mask (numpy bool) are the
    training-time masks (None = eval).  Returns (log_probs [B,T,V], pad_mask [B,T])."""
    p = "encoder."
    fx = conv_feature_extractor(x, sd, p + "feature_extractor.", conv_features=conv_features).transpose(1, 2)
    feats = F.layer_norm(fx, (fx.shape[-1],), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], 1e-5)
    T = feats.shape[1]
    if pad_mask is not None:
        pad_mask = downsample_pad_mask(pad_mask, T)
    feats = F.linear(feats, sd[p + "proj_to_input.layer.weight"], sd[p + "proj_to_input.layer.bias"])
    if time_mask is not None:
        feats = torch.where(torch.from_numpy(time_mask).unsqueeze(-1), sd[p + "mask_emb"].view(1, 1, -1), feats)
    if channel_mask is not None:
        feats = feats.masked_fill(torch.from_numpy(channel_mask).unsqueeze(1), 0.0)               # :719-721
    enc = audio_transformer_encoder(feats, sd, num_layers, num_heads, pad_mask, p + "encoder.", ln_eps, posconv_pad,
                                    training=training, layer_drop=layer_drop,
                                    layerdrop_rng_draws=layerdrop_rng_draws)
    logits = F.linear(enc, sd["proj.weight"], sd["proj.bias"])                                    # :769
    return F.log_softmax(logits, dim=-1), pad_mask                                                # :770


def sequence_mask(lengths: torch.Tensor, max_len: int) -> torch.Tensor:
    """eight_mile.pytorch.layers.sequence_mask: bool [B, max_len], True where index < length."""
    return torch.arange(max_len).unsqueeze(0) < lengths.unsqueeze(1)


# --------------------------------------------------------------------------------------
# a11: CTC loss                                               audio8/ctc.py:186-206
# --------------------------------------------------------------------------------------
PAD, EOS, GO = 1, 2, 0     # audio8/train.py:22-23 (+ eight_mile default EOS=2)


def ctc_strip_targets(targets: np.ndarray):
    """targets int [B,S] padded with PAD(1)/EOS(2) -> 1-D concatenation of the real labels (:193-194)."""
    t = np.asarray(targets)
    return t[(t != PAD) & (t != EOS)]


def ctc_loss_np(log_prob, input_lengths, targets, target_lengths, reduction="sum", zero_infinity=True, blank=GO):
    """Own alpha/beta log-space recurrences (float64) following the published CTC algorithm as
    ATen's native ctc_loss computes it (the reference disables cuDNN, audio8/ctc.py:196).
    log_prob [T,B,V]; returns (loss scalar, grad [T,B,V]) with ATen's semantics:
    reduction 'mean' divides each utterance's nll by clamp(target_len,1) then averages over B;
    zero_infinity zeroes both the infinite loss and its gradient.  Small sizes only."""
    lp = np.asarray(log_prob, dtype=np.float64)
    T, B, V = lp.shape
    flat = ctc_strip_targets(targets)
    il = np.asarray(input_lengths).astype(int)
    tl = np.asarray(target_lengths).astype(int)
    NEG = -np.inf
    nll = np.zeros(B)
    grad = np.zeros_like(lp)
    off = 0
    for b in range(B):
        S, Tb = tl[b], il[b]
        lab = flat[off:off + S]
        off += S
        L = 2 * S + 1
        ext = np.full(L, blank, dtype=int)
        ext[1::2] = lab
        can_skip = np.zeros(L, dtype=bool)
        can_skip[3::2] = lab[1:] != lab[:-1]
        if Tb == 0:                                   # no frames: only the empty target has an alignment (ATen: loss 0), else -inf likelihood
            nll[b] = 0.0 if S == 0 else np.inf
            continue
        la = np.full((Tb, L), NEG)
        la[0, 0] = lp[0, b, blank]
        if L > 1:
            la[0, 1] = lp[0, b, ext[1]]
        for t in range(1, Tb):
            prev = la[t - 1]
            a1 = np.concatenate([[NEG], prev])[:L]
            a2 = np.where(can_skip, np.concatenate([[NEG, NEG], prev])[:L], NEG)           # (slices: L = 1, the empty target, included)
            la[t] = np.logaddexp(np.logaddexp(prev, a1), a2) + lp[t, b, ext]
        ll = np.logaddexp(la[Tb - 1, L - 1], la[Tb - 1, L - 2]) if L > 1 else la[Tb - 1, 0]
        nll[b] = -ll
        if not np.isfinite(ll):
            continue
        lb = np.full((Tb, L), NEG)
        lb[Tb - 1, L - 1] = lp[Tb - 1, b, blank]
        if L > 1:
            lb[Tb - 1, L - 2] = lp[Tb - 1, b, ext[L - 2]]
        skip_next = np.concatenate([can_skip[2:], [False, False]])[:L]
        for t in range(Tb - 2, -1, -1):
            nxt = lb[t + 1]
            b1 = np.concatenate([nxt[1:], [NEG]])
            b2 = np.where(skip_next, np.concatenate([nxt[2:], [NEG, NEG]])[:L], NEG)
            lb[t] = np.logaddexp(np.logaddexp(nxt, b1), b2) + lp[t, b, ext]
        ab = la + lb                                        # log(alpha*beta), contains lp once extra
        for t in range(Tb):
            acc = np.full(V, NEG)
            np.logaddexp.at(acc, ext, ab[t])
            # d nll / d lp[t,b,c] = exp(lp) - exp(acc - lp + nll)   (ATen ctc_loss_backward)
            grad[t, b] = np.exp(lp[t, b]) - np.exp(acc - lp[t, b] - ll)
    inf = ~np.isfinite(nll)
    if zero_infinity:
        nll[inf] = 0.0
        grad[:, inf] = 0.0
    if reduction == "sum":
        return nll.sum(), grad
    if reduction == "mean":
        denom = np.maximum(tl, 1)
        return (nll / denom).mean(), grad / denom[None, :, None] / B
    return nll, grad


# --------------------------------------------------------------------------------------
# a12: greedy CTC metrics                                     audio8/ctc.py:139-183, 63-66
# --------------------------------------------------------------------------------------
def edit_distance(a, b) -> int:
    """Levenshtein distance over two sequences (what `editdistance.eval` returns)."""
    a, b = list(a), list(b)
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def postproc_letters(sentence):
    return ''.join(sentence).replace(" ", "").replace("|", " ").strip()      # audio8/ctc.py:63-66


def ctc_metrics(lprobs_t, target, input_lengths, index2vocab, postproc_fn=postproc_letters):
    """lprobs_t [B,T,V] (cpu), target int [B,S] -> dict(wv_errors,w_errors,w_total,c_errors,c_total).
    Quirk kept: slices rows by the *sample* length (:156)."""
    c_err = c_len = w_errs = w_len = 0
    lp = torch.as_tensor(lprobs_t)
    tg = torch.as_tensor(target)
    for b in range(lp.shape[0]):
        row = lp[b][: int(input_lengths[b])]
        t = tg[b]
        targ = t[(t != PAD) & (t != EOS)].tolist()
        toks = row.argmax(dim=-1).unique_consecutive()
        pred = toks[toks != GO].tolist()
        c_err += edit_distance(pred, targ)
        c_len += len(targ)
        tw = postproc_fn([index2vocab[i] for i in targ]).split()
        pw = postproc_fn([index2vocab[i] for i in pred]).split()
        w_errs += edit_distance(pw, tw)
        w_len += len(tw)
    return dict(wv_errors=w_errs, w_errors=w_errs, w_total=w_len, c_errors=c_err, c_total=c_len)


# --------------------------------------------------------------------------------------
# deterministic random state_dict with the reference's key names and init   
# --------------------------------------------------------------------------------------
def init_state_dict(seed=0, d_model=768, num_heads=12, num_layers=12, d_ff=None, final_dim=256, num_vq_vars=320,
                    num_vq_groups=2, conv_features=CONV_FEATURES[16], ctc_labels: Optional[int] = None, pdrop=0.1):
    """Random-init weights of the architecture (numpy RandomState -> stable across torch versions).
    Distributions follow the reference init where it matters for conditioning (kaiming convs,
    xavier linears, uniform(0,1) vars/mask_emb, N(0,1) VQ projection, uniform(+-std) pos-conv)."""
    rs = np.random.RandomState(seed)
    d_ff = d_ff or 4 * d_model
    sd = {}
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    cin = 1
    for i, (co, k, s) in enumerate(conv_features):
        bound = math.sqrt(6.0 / (cin * k))
        sd[f"feature_extractor.conv_layers.{i}.0.weight"] = f32(rs.uniform(-bound, bound, (co, cin, k)))
        cin = co
    sd["feature_extractor.conv_layers.0.2.weight"] = f32(np.ones(cin))
    sd["feature_extractor.conv_layers.0.2.bias"] = f32(np.zeros(cin))
    sd["layer_norm.weight"] = f32(np.ones(cin))
    sd["layer_norm.bias"] = f32(np.zeros(cin))

    def dense(name, n_in, n_out):
        b = math.sqrt(6.0 / (n_in + n_out))
        sd[f"{name}.layer.weight"] = f32(rs.uniform(-b, b, (n_out, n_in)))
        sd[f"{name}.layer.bias"] = f32(np.zeros(n_out))

    dense("proj_to_input", cin, d_model)
    sd["mask_emb"] = f32(rs.uniform(0, 1, d_model))
    std = math.sqrt(4 * (1.0 - pdrop) / (128 * d_model))
    v = rs.uniform(-std, std, (d_model, d_model // 16, 128))
    sd["encoder.pos_conv.conv.1.weight_v"] = f32(v)
    sd["encoder.pos_conv.conv.1.weight_g"] = f32(np.sqrt((v ** 2).sum(axis=(0, 1), keepdims=True)))
    sd["encoder.pos_conv.conv.1.bias"] = f32(np.zeros(d_model))
    sd["encoder.ln.weight"] = f32(np.ones(d_model))
    sd["encoder.ln.bias"] = f32(np.zeros(d_model))
    for i in range(num_layers):
        p = f"encoder.transformer.encoders.{i}."
        for n in ("w_Q", "w_K", "w_V", "w_O"):
            dense(p + "self_attn." + n, d_model, d_model)
        dense(p + "ffn.0", d_model, d_ff)
        dense(p + "ffn.3", d_ff, d_model)
        for ln in ("ln1", "ln2"):
            sd[p + ln + ".weight"] = f32(np.ones(d_model))
            sd[p + ln + ".bias"] = f32(np.zeros(d_model))
    if ctc_labels is None:
        sd["quantizer.vars"] = f32(rs.uniform(0, 1, (1, num_vq_groups * num_vq_vars, final_dim // num_vq_groups)))
        sd["quantizer.weight_proj.weight"] = f32(rs.standard_normal((num_vq_groups * num_vq_vars, cin)))
        sd["quantizer.weight_proj.bias"] = f32(np.zeros(num_vq_groups * num_vq_vars))
        dense("project_q", final_dim, final_dim)
        dense("final_proj", d_model, final_dim)
    else:
        sd = {"encoder." + k: v for k, v in sd.items()}
        b = math.sqrt(6.0 / (d_model + ctc_labels))
        sd["proj.weight"] = f32(rs.uniform(-b, b, (ctc_labels, d_model)))
        sd["proj.bias"] = f32(np.zeros(ctc_labels))
    return sd


def crop_to_min_batches(samples, max_length, target_tokens_per_batch):
    """AudioFileDataset.__iter__ (audio8/data.py:409-426) restated: accumulate samples while len(samples) * min_length is
    below the budget; once it is not, yield the accumulated samples cropped to min_length and DROP the sample in hand
    (the reference only appends inside the `if`); min_length carries over to the next batch."""
    out = []
    min_length, predicted, acc = max_length, 0, []
    for smp in samples:
        if predicted < target_tokens_per_batch:
            min_length = min(min_length, len(smp))
            acc.append(smp)
            predicted = len(acc) * min_length
        else:
            out.append(np.stack([x[:min_length] for x in acc]))
            acc, predicted = [], 0
    return out
