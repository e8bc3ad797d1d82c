#!/usr/bin/env python3
"""Generate golden fixtures under tests/golden/ from the reference's own executable pieces.

Runs ONLY in the build container (needs /root/reference).  The reference module
`audio8/wav2vec2.py` cannot be imported (its `eight_mile` dependency is absent),
so this script parses the file with `ast` and exec()s, in a scratch namespace, only the
top-level definitions that do not need eight_mile:

    create_mask            audio8/wav2vec2.py:189-216
    Sampler                audio8/wav2vec2.py:955-976
    GumbelVectorQuantizer  audio8/wav2vec2.py:459-576
    Wav2Vec2Loss           audio8/wav2vec2.py:371-392
    ConvFeatureExtractionModel  audio8/wav2vec2.py:399-456  (needs one injected name,
        `pytorch_conv1d`, whose only job in eight_mile is nn.Conv1d + init; every conv
        weight is overwritten with the fixture's own weights before use)
    CTCLoss                audio8/ctc.py:186-206 (needs Offsets.{GO,PAD,EOS} = 0,1,2, the
        values audio8/train.py:22-23 installs)
    ctc_metrics, postproc_*  audio8/ctc.py:139-183, 63-72 (G8; `editdistance` supplied as a plain DP)
    key maps, convert_keys   audio8/wav2vec2.py:39-168 (G9)
G7 (end-to-end cfg1) comes from the CPU restatement, not from the reference (see gen_e2e).

No reference source text or bytecode is written anywhere: the outputs are .npz files of
inputs and expected outputs only (tests/golden/*.npz).

Usage:  python tools/gen_golden.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import ast
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


WANTED_W2V = {
    "create_mask", "Sampler", "GumbelVectorQuantizer", "Wav2Vec2Loss", "ConvFeatureExtractionModel",
    "CONV_FEATURES", "START_TEMP", "END_TEMP", "TEMP_DECAY_FACTOR", "XE_WGT", "DIVERSITY_WGT",
}
WANTED_CTC = {"CTCLoss", "ctc_metrics", "postproc_letters", "postproc_bpe"}
WANTED_MAPS = {"CheckpointMapping", "W2V_CTC_NESTED_MAP", "W2V2_CTC_FLAT_MAP_16", "W2V2_CTC_FLAT_MAP_8", "W2V_NESTED_MAP", "W2V2_FLAT_MAP",
               "W2V_MAP", "W2V_CTC_MAP_16", "W2V_CTC_MAP_8", "W2V_CTC_MAP", "convert_keys"}


def _extract(path, wanted, ns):
    src = open(path).read()
    tree = ast.parse(src)
    body = []
    for node in tree.body:
        name = None
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)):
            name = node.name
        elif isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            name = node.targets[0].id
        if name in wanted:
            body.append(node)
    mod = ast.Module(body=body, type_ignores=[])
    exec(compile(mod, path, "exec"), ns)
    return ns


def load_reference(ref_root):
    from typing import Tuple, List, Optional, Dict
    import contextlib

    class Offsets:
        GO, PAD, EOS, UNK = 0, 1, 2, 3

    def pytorch_conv1d(in_channels, out_channels, fsz, unif=0, padding=0, initializer=None, stride=1, bias=True, groups=1):
        return nn.Conv1d(in_channels, out_channels, fsz, padding=padding, stride=stride, bias=bias, groups=groups)

    from collections import namedtuple
    ns = dict(torch=torch, nn=nn, F=F, np=np, math=math, Tuple=Tuple, List=List, Optional=Optional, Dict=Dict,
              contextlib=contextlib, Offsets=Offsets, pytorch_conv1d=pytorch_conv1d, namedtuple=namedtuple)
    _extract(os.path.join(ref_root, "audio8", "wav2vec2.py"), WANTED_W2V | WANTED_MAPS, ns)
    _extract(os.path.join(ref_root, "audio8", "ctc.py"), WANTED_CTC, ns)
    return types.SimpleNamespace(**ns)


def t2n(t):
    return t.detach().cpu().numpy()


def gen_masks(ref, out):
    """G1: create_mask, bit-exact bool arrays."""
    d = {}
    cases = []
    for seed in (0, 1, 2):
        for shape in ((2, 49), (8, 781), (4, 999), (16, 624)):
            for (p, ml) in ((0.65, 10), (0.5, 10)):
                cases.append((seed, shape, p, ml))
        # channel mask case used by the fine-tune encoder: C = 768, p = 0.1, len 64
        cases.append((seed, (16, 768), 0.1, 64))
        cases.append((seed, (3, 12), 0.65, 10))   # tiny: exercises sz - min_len <= num_mask branch
    keys = []
    for (seed, shape, p, ml) in cases:
        np.random.seed(seed)
        m = ref.create_mask(shape, p_start=p, mask_length=ml)
        tail = np.random.rand()  # RNG position after the call (pins the number of draws)
        key = f"s{seed}_b{shape[0]}_t{shape[1]}_p{int(p * 100)}_l{ml}"
        keys.append(key)
        d[key] = np.packbits(m, axis=None)
        d[key + "_shape"] = np.array(shape)
        d[key + "_tail"] = np.array(tail)
    d["keys"] = np.array(keys)
    np.savez_compressed(os.path.join(out, "g1_masks.npz"), **d)


def gen_negatives(ref, out):
    """G2: Sampler.negatives indices + combined RNG-order case."""
    d = {}
    keys = []
    for seed in (0, 1, 2):
        for (B, Tm, N) in ((2, 21, 100), (8, 368, 100), (3, 5, 7)):
            if seed > 0 and Tm == 368:
                continue
            np.random.seed(seed)
            s = ref.Sampler(N)
            y = torch.zeros(B, Tm, 4)
            _, idx = s.negatives(y)
            key = f"s{seed}_b{B}_t{Tm}_n{N}"
            keys.append(key)
            d[key] = t2n(idx).astype(np.int32)
    # RNG order: seed; create_mask; negatives
    np.random.seed(7)
    m = ref.create_mask((2, 49), p_start=0.65, mask_length=10)
    Tm = int(m[0].sum())
    s = ref.Sampler(100)
    _, idx = s.negatives(torch.zeros(2, Tm, 4))
    d["order_mask"] = m
    d["order_idx"] = t2n(idx).astype(np.int32)
    d["keys"] = np.array(keys)
    np.savez_compressed(os.path.join(out, "g2_negatives.npz"), **d)


def conv_fixture_weights(seed):
    """Deterministic conv-stack weights from a numpy RandomState (stable across torch versions).
    Returns (list of 7 conv weights [Cout,Cin,k], gn_weight[512], gn_bias[512]) as float32 arrays."""
    rs = np.random.RandomState(seed)
    feats = [(512, 10, 5), (512, 3, 2), (512, 3, 2), (512, 3, 2), (512, 3, 2), (512, 2, 2), (512, 2, 2)]
    ws, cin = [], 1
    for (co, k, s) in feats:
        ws.append((rs.standard_normal((co, cin, k)) * math.sqrt(2.0 / (cin * k))).astype(np.float32))
        cin = co
    gn_w = (1.0 + 0.1 * rs.standard_normal(512)).astype(np.float32)
    gn_b = (0.1 * rs.standard_normal(512)).astype(np.float32)
    return ws, gn_w, gn_b


def gen_conv(ref, out):
    """G3: conv feature extractor forward + gradients (fp32).  Weights come from
    conv_fixture_weights(seed) (a recipe, not stored); big weight-grads are stored subsampled
    (every 97th element of the flattened array) together with their L2 norm."""
    d = {}
    for tag, (B, L) in (("a", (2, 16000)), ("b", (1, 4000))):
        seed = 100 + B
        ws, gn_w, gn_b = conv_fixture_weights(seed)
        fx = ref.ConvFeatureExtractionModel(ref.CONV_FEATURES[16])
        with torch.no_grad():
            for i, blk in enumerate(fx.conv_layers):
                blk[0].weight.copy_(torch.from_numpy(ws[i]))
            fx.conv_layers[0][2].weight.copy_(torch.from_numpy(gn_w))
            fx.conv_layers[0][2].bias.copy_(torch.from_numpy(gn_b))
        rs = np.random.RandomState(seed + 1000)
        x = torch.from_numpy(np.clip(0.1 * rs.standard_normal((B, L)), -1, 1).astype(np.float32))
        x.requires_grad_(True)
        y = fx(x)
        g = torch.from_numpy((rs.standard_normal(tuple(y.shape)) / math.sqrt(y.shape[-1])).astype(np.float32))
        y.backward(g)
        d[f"{tag}_seed"] = np.array(seed)
        d[f"{tag}_x"] = t2n(x)
        d[f"{tag}_y"] = t2n(y)
        d[f"{tag}_gy"] = t2n(g)
        d[f"{tag}_gx"] = t2n(x.grad)
        h = x.detach().unsqueeze(1)
        stats = []
        for blk in fx.conv_layers:
            h = blk(h)
            stats.append([h.double().mean().item(), h.double().abs().mean().item(), float(h.shape[-1])])
        d[f"{tag}_layer_stats"] = np.array(stats)
        for i, blk in enumerate(fx.conv_layers):
            gw = t2n(blk[0].weight.grad).reshape(-1)
            d[f"{tag}_gw{i}_sub"] = gw[::97].copy() if gw.size > 100000 else gw.copy()
            d[f"{tag}_gw{i}_norm"] = np.array(np.sqrt((gw.astype(np.float64) ** 2).sum()))
        d[f"{tag}_g_gn_w"] = t2n(fx.conv_layers[0][2].weight.grad)
        d[f"{tag}_g_gn_b"] = t2n(fx.conv_layers[0][2].bias.grad)
    np.savez_compressed(os.path.join(out, "g3_conv.npz"), **d)


def gen_vq(ref, out):
    """G4: Gumbel VQ, eval (argmax) and train (injected Gumbel noise)."""
    d = {}
    B, Tm, C, V, G, Q = 2, 21, 512, 320, 2, 256
    torch.manual_seed(11)
    vq = ref.GumbelVectorQuantizer(C, V, ref.START_TEMP, ref.END_TEMP, ref.TEMP_DECAY_FACTOR, G, Q)
    d["tau"] = np.array(vq.curr_temperature, dtype=np.float64)
    d["tau_after_set_num_updates_1000"] = np.array(0.0)
    y = torch.randn(B, Tm, C) * 0.5
    d["y"] = t2n(y)
    d["vars"] = t2n(vq.vars)
    d["wp"] = t2n(vq.weight_proj.weight * 0.05)  # scaled so that logits are O(1)
    d["bp"] = t2n(vq.weight_proj.bias)
    with torch.no_grad():
        vq.weight_proj.weight.mul_(0.05)
    # eval
    vq.eval()
    yy = y.clone().requires_grad_(True)
    q, ppl = vq(yy)
    d["eval_q"] = t2n(q)
    d["eval_ppl"] = t2n(ppl)
    # train with injected noise: F.gumbel_softmax draws
    #   gumbels = -torch.empty_like(logits).exponential_().log()   (torch/nn/functional.py gumbel_softmax)
    vq.train()
    vq.zero_grad()
    torch.manual_seed(123)
    logits_shape = (B * Tm * G, V)
    gum = -torch.empty(logits_shape).exponential_().log()
    d["gumbel"] = t2n(gum)
    torch.manual_seed(123)
    yy = y.clone().requires_grad_(True)
    q, ppl = vq(yy)
    gq = torch.randn_like(q)
    (q * gq).sum().backward(retain_graph=True)
    d["train_q"] = t2n(q)
    d["train_ppl"] = t2n(ppl)
    d["train_gq"] = t2n(gq)
    d["train_gy"] = t2n(yy.grad)
    d["train_gvars"] = t2n(vq.vars.grad)
    d["train_gwp"] = t2n(vq.weight_proj.weight.grad)
    d["train_gbp"] = t2n(vq.weight_proj.bias.grad)
    # gradient of prob_ppl alone
    vq.zero_grad()
    torch.manual_seed(123)
    yy2 = y.clone().requires_grad_(True)
    q2, ppl2 = vq(yy2)
    ppl2.backward()
    d["ppl_gy"] = t2n(yy2.grad)
    vq.set_num_updates(1000)
    d["tau_after_set_num_updates_1000"] = np.array(vq.curr_temperature, dtype=np.float64)
    np.savez_compressed(os.path.join(out, "g4_vq.npz"), **d)


def gen_loss(ref, out):
    """G5: contrastive + diversity loss on fixed model outputs."""
    d = {}
    B, T, Tm, C, N = 2, 49, 21, 256, 100

    class FakeModel:
        def __call__(self, features):
            return self.out

    np.random.seed(5)
    torch.manual_seed(5)
    mask = np.zeros((B, T), dtype=bool)
    for b in range(B):
        mask[b, np.random.choice(T, Tm, replace=False)] = True
    outputs = torch.randn(B, T, C, requires_grad=True)
    latents = torch.randn(B, Tm, C, requires_grad=True)
    gs = torch.tensor(311.5, requires_grad=True)
    fm = FakeModel()
    fm.out = (outputs, latents, gs, torch.from_numpy(mask))
    loss_fn = ref.Wav2Vec2Loss(640, N)
    np.random.seed(9)
    loss = loss_fn(fm, None)
    loss.backward()
    np.random.seed(9)
    _, idx = ref.Sampler(N).negatives(latents.detach())
    # logits as the reference builds them (for the logits pin)
    with torch.no_grad():
        y = latents.unsqueeze(0)
        o = outputs[torch.from_numpy(mask).unsqueeze(-1).expand_as(outputs)].view(B, -1, C).unsqueeze(0)
        neg = latents.view(-1, C)[idx.view(-1)].view(B, Tm, N, C).permute(2, 0, 1, 3)
        logits = torch.cosine_similarity(o, torch.cat([y, neg], 0), dim=-1).transpose(2, 0).reshape(-1, N + 1)
    d.update(mask=mask, outputs=t2n(outputs), latents=t2n(latents), gs=t2n(gs), idx=t2n(idx).astype(np.int32),
             logits=t2n(logits), loss=t2n(loss), g_outputs=t2n(outputs.grad), g_latents=t2n(latents.grad),
             g_gs=t2n(gs.grad), n_vars=np.array(640))
    np.savez_compressed(os.path.join(out, "g5_loss.npz"), **d)


def gen_ctc(ref, out):
    """G6: CTCLoss (both reductions, PAD/EOS-padded targets, an infeasible row)."""
    d = {}
    for tag, (T, B, V, Smax) in (("a", (49, 2, 32, 12)), ("b", (624, 4, 32, 300))):
        torch.manual_seed(20 + T)
        lp = torch.randn(T, B, V).log_softmax(-1)
        tl = torch.randint(max(1, Smax // 3), Smax + 1, (B,))
        tl[0] = Smax
        il = torch.randint(T // 2, T + 1, (B,))
        il[0] = T
        if tag == "a":
            il[1] = 5          # infeasible: T < S  -> inf -> zero_infinity
            tl[1] = 9
        targets = torch.full((B, Smax + 2), 1, dtype=torch.long)  # PAD
        for b in range(B):
            targets[b, : tl[b]] = torch.randint(4, V, (int(tl[b]),))
            targets[b, tl[b]] = 2  # EOS
        d[f"{tag}_lp"] = t2n(lp)
        d[f"{tag}_il"] = t2n(il)
        d[f"{tag}_tl"] = t2n(tl)
        d[f"{tag}_targets"] = t2n(targets)
        for red in ("sum", "mean"):
            x = lp.clone().requires_grad_(True)
            loss = ref.CTCLoss(reduction_type=red)(x, il, targets, tl)
            loss.backward()
            d[f"{tag}_{red}_loss"] = t2n(loss)
            d[f"{tag}_{red}_grad"] = t2n(x.grad)
    np.savez_compressed(os.path.join(out, "g6_ctc.npz"), **d)


def _own_levenshtein(a, b):
    a, b = list(a), list(b)
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def gen_e2e(out):
    """G7: one cfg1 pretrain step (wav2vec2-base, 12 layers, 2 x 16 000 samples) through the CPU restatement (oracle/ref_cpu.py).
    NOT produced by the reference (its model cannot be built without eight_mile): pins the restatement against drift;
    the eight_mile-owned semantics inside it stay "parity unpinned"."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle import ref_cpu as R
    sd = {k: v.clone().requires_grad_(True) for k, v in R.init_state_dict(seed=0, num_layers=12).items()}
    x = (0.1 * torch.randn(2, 16000, generator=torch.Generator().manual_seed(1234))).clamp_(-1, 1)
    np.random.seed(0)
    T = R.conv_out_lengths(16000)[-1]
    Tm = int(R.create_mask((2, T), 0.65, 10)[0].sum())
    gum = -torch.empty(2 * Tm * 2, 320).exponential_(generator=torch.Generator().manual_seed(5)).log()
    np.random.seed(0)
    loss, aux = R.pretrain_loss(x, sd, num_layers=12, training=True, gumbel=gum)
    loss.backward()
    d = dict(loss=t2n(loss), ppl=t2n(aux["ppl"]), time_mask=aux["time_mask"], neg_idx=aux["neg_idx"].astype(np.int32),
             ctx_sub=t2n(aux["ctx"]).reshape(-1)[::53].copy(), q_sub=t2n(aux["q"]).reshape(-1)[::53].copy(),
             state_seed=np.array(0), x_seed=np.array(1234), gumbel_seed=np.array(5), numpy_seed=np.array(0))
    names = sorted(sd)
    d["grad_names"] = np.array(names)
    d["grad_norms"] = np.array([float(sd[k].grad.double().norm()) for k in names])
    d["grad_sub_final_proj_w"] = t2n(sd["final_proj.layer.weight"].grad).reshape(-1)[::97].copy()
    d["grad_sub_conv0_w"] = t2n(sd["feature_extractor.conv_layers.0.0.weight"].grad).reshape(-1).copy()
    np.savez_compressed(os.path.join(out, "g7_e2e_cfg1.npz"), **d)


def gen_metrics(ref, out):
    """G8: ctc_metrics greedy path (audio8/ctc.py:139-183) on a tiny letter vocabulary.  `editdistance` (third-party, absent)
    is supplied by this script as a plain Levenshtein DP; what the fixture pins is the reference's decode / strip /
    post-processing / word-splitting around it, including the slice by SAMPLE length (`lp[:inp_l]`)."""
    fake = types.ModuleType("editdistance")
    fake.eval = _own_levenshtein
    sys.modules["editdistance"] = fake
    try:
        vocab = ['<s>', '<pad>', '</s>', '<unk>', '|'] + [chr(97 + i) for i in range(27)]
        i2v = {i: c for i, c in enumerate(vocab)}
        g = torch.Generator().manual_seed(8)
        B, T, V = 6, 40, 32
        lp = torch.randn(B, T, V, generator=g)
        # make decodes look like text: long runs of a few symbols, blanks and word separators
        path = torch.randint(0, V, (B, T // 4), generator=g).repeat_interleave(4, 1)
        path[:, ::7] = 0
        path[:, 5::9] = 4
        lp.scatter_(2, path.unsqueeze(-1), 9.0)
        lp = lp.log_softmax(-1)
        tl = torch.tensor([9, 5, 0, 12, 3, 7])
        targets = torch.full((B, 14), 1, dtype=torch.long)
        for b in range(B):
            targets[b, : tl[b]] = torch.randint(4, V, (int(tl[b]),), generator=g)
            if b % 2 == 0:
                targets[b, 2:tl[b]:3] = 4
            targets[b, tl[b]] = 2
        d = dict(lp=t2n(lp), targets=t2n(targets), vocab=np.array(vocab))
        for tag, il in (("frames", torch.tensor([40, 33, 20, 40, 7, 1])), ("samples", torch.tensor([12800, 10560, 6400, 12800, 2240, 320]))):
            for pn, fn in (("letters", ref.postproc_letters), ("bpe", ref.postproc_bpe)):
                m = ref.ctc_metrics(lp, targets, il, i2v, postproc_fn=fn)
                d[f"{tag}_{pn}_il"] = t2n(il)
                d[f"{tag}_{pn}"] = np.array([m["wv_errors"], m["w_errors"], m["w_total"], m["c_errors"], m["c_total"]])
        np.savez_compressed(os.path.join(out, "g8_ctc_metrics.npz"), **d)
    finally:
        del sys.modules["editdistance"]


def gen_keymaps(ref, out):
    """G9: the reference's fairseq -> audio8 key maps (audio8/wav2vec2.py:39-151) as (source, destination) string tables, and
    convert_keys (audio8/wav2vec2.py:154-168) run on synthetic fairseq-named dicts: the output key ORDER and which value
    lands under which key (values are tagged integers)."""
    d = {}
    for name in ("W2V_CTC_NESTED_MAP", "W2V2_CTC_FLAT_MAP_16", "W2V2_CTC_FLAT_MAP_8", "W2V_NESTED_MAP", "W2V2_FLAT_MAP"):
        m = getattr(ref, name)
        d[name + "_src"] = np.array(list(m.keys()))
        d[name + "_dst"] = np.array(list(m.values()))
    for tag, cm, nl in (("pretrain", ref.W2V_MAP, 3), ("ctc16", ref.W2V_CTC_MAP[16], 2), ("ctc8", ref.W2V_CTC_MAP[8], 2)):
        src = {}
        for i in range(nl):
            for k in cm.nested:
                src[k.format(i)] = len(src)
        for k in cm.flat:
            src[k] = len(src)
        src["some.unmapped.key"] = len(src)
        src["another_unmapped"] = len(src)
        d[f"{tag}_in_keys"] = np.array(list(src.keys()))
        outd = ref.convert_keys(nl, dict(src), cm.nested, cm.flat)
        d[f"{tag}_out_keys"] = np.array(list(outd.keys()))
        d[f"{tag}_out_vals"] = np.array(list(outd.values()))
        d[f"{tag}_layers"] = np.array(nl)
    np.savez_compressed(os.path.join(out, "g9_keymaps.npz"), **d)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    torch.set_num_threads(4)
    ref = load_reference(args.ref)
    gen_masks(ref, args.out)
    gen_negatives(ref, args.out)
    gen_conv(ref, args.out)
    gen_vq(ref, args.out)
    gen_loss(ref, args.out)
    gen_ctc(ref, args.out)
    gen_e2e(args.out)
    gen_metrics(ref, args.out)
    gen_keymaps(ref, args.out)
    for f in sorted(os.listdir(args.out)):
        print(f, os.path.getsize(os.path.join(args.out, f)))


if __name__ == "__main__":
    sys.exit(main())
