"""Branches of the path that both reference drivers expose and that had never executed: layer drop
(audio8/pretrain.py:47, train.py:77 -> audio8/wav2vec2.py:613-622), the fine-tune model at full depth with ragged
pad masks (audio8/wav2vec2.py:696-723,765-770) and the 8 kHz feature extractor (CONV_FEATURES[8], six conv layers,
audio8/wav2vec2.py:26-29).  All f32, HIP path through the C ABI against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_model_gpu import rel_err, synth, restore_dtype  # noqa: F401  (fixture + helpers)

pytestmark = pytest.mark.gpu


def _grad_check(got, sdg, tol, skip=()):
    worst, worst_k = 0.0, None
    for k, ref in sdg.items():
        if any(k.startswith(s) for s in skip):
            continue
        assert got[k].grad is not None and ref.grad is not None, k
        if k.endswith("w_K.layer.bias"):      # exactly zero in exact arithmetic (softmax shift invariance): noise on both sides
            qn = got[k.replace("w_K", "w_Q")].grad.norm().item()
            assert got[k].grad.norm().item() < 1e-2 * qn + 1e-9 and ref.grad.norm().item() < 1e-2 * qn + 1e-9, k
            continue
        e = rel_err(got[k].grad, ref.grad)
        if e > worst:
            worst, worst_k = e, k
        assert e < tol, f"{k}: rel grad err {e}"
    print("worst relative gradient error:", worst, worst_k)


def test_layer_drop_parity_and_untouched_parameters_f32(restore_dtype):
    """layer_drop = 0.5 on a 4-layer model: the product must drop the layers the oracle drops for the same numpy seed (the draws sit
    between the span mask and the negatives in the global RNG order), match its loss and gradients, leave the dropped layers'
    parameters without a gradient, and Stepper.step() must not touch them (no decay, no moments, own step count)."""
    from audio8_amd import ops, wav2vec2 as W
    from audio8_amd.trainer import Stepper
    from oracle import ref_cpu as R
    ops.set_compute_dtype(torch.float32)
    NL, B, L, p_drop = 4, 2, 16000, 0.5
    T = R.conv_out_lengths(L)[-1]
    # a seed whose four draws drop some layers but not all (the draws follow create_mask in the RNG stream)
    seed = None
    for cand in range(50):
        np.random.seed(cand)
        R.create_mask((B, T), 0.65, 10)
        dropped = [bool(np.random.random() < p_drop) for _ in range(NL)]
        if 0 < sum(dropped) < NL and not dropped[-1]:
            seed = cand
            break
    assert seed is not None
    print("seed", seed, "drops layers", [i for i, d in enumerate(dropped) if d])
    sd = R.init_state_dict(seed=seed, num_layers=NL)
    x = synth(B, L)
    np.random.seed(seed)
    Tm = int(R.create_mask((B, T), 0.65, 10)[0].sum())
    gum = -torch.empty(B * Tm * 2, 320).exponential_(generator=torch.Generator().manual_seed(5)).log()
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    np.random.seed(seed)
    loss_ref, aux = R.pretrain_loss(x, sdg, num_layers=NL, training=True, gumbel=gum, layer_drop=p_drop)
    loss_ref.backward()
    rng_after_ref = np.random.random()

    model = W.create_model(num_layers=NL, dropout=0.0, dropout_input=0.0, dropout_features=0.0, layer_drop=p_drop)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().train()
    model.quantizer.gumbel_override = gum.cuda()
    st = Stepper(model, lr=1e-3, weight_decay=1e-2, clip=1.0)
    try:
        loss_fn = W.create_loss(640, 100)
        st.zero_grad()
        np.random.seed(seed)
        loss = loss_fn(model, x.cuda())
        loss.backward()
        assert np.random.random() == rng_after_ref                # same number of draws, same order
        torch.cuda.synchronize()
        np.testing.assert_allclose(loss.item(), loss_ref.item(), rtol=1e-4)
        got = dict(model.named_parameters())
        dropped_pfx = tuple(f"encoder.transformer.encoders.{i}." for i, d in enumerate(dropped) if d)
        for k, ref in sdg.items():
            if k.startswith(dropped_pfx):
                assert ref.grad is None or float(ref.grad.abs().max()) == 0.0, k       # the oracle dropped the same layers ...
                assert float(got[k].grad.abs().max()) == 0.0, k                          # ... and the product wrote nothing there
        _grad_check(got, sdg, 1e-4, skip=dropped_pfx)
        names = {id(p): n for n, p in model.named_parameters()}
        touched = {names[id(p)]: t for p, t in zip(st.state.params, st.reducer.touched)}
        assert all(touched[k] != k.startswith(dropped_pfx) for k in sdg), [k for k in sdg if touched[k] == k.startswith(dropped_pfx)]
        before = {k: p.detach().clone() for k, p in got.items()}
        st.step()
        torch.cuda.synchronize()
        for i, p in enumerate(st.state.params):
            k = names[id(p)]
            moved = float((p.detach() - before[k]).abs().max())
            if k.startswith(dropped_pfx):
                assert moved == 0.0 and st.param_steps[i] == 0, (k, moved)     # untouched: no weight decay either (torch AdamW skips grad None)
                o = st.state.offsets[i]
                assert float(st.state.flat_m[o:o + p.numel()].abs().max()) == 0.0
            else:
                assert st.param_steps[i] == 1, k
                assert moved > 0.0 or p.numel() == 0, k
    finally:
        st.close()


def test_ctc_finetune_parity_f32_full_depth_ragged(restore_dtype):
    """The fine-tune model at wav2vec2-BASE depth (12 layers) with ragged utterance lengths (key-padding masks in every attention
    layer, pad-mask downsampling, time + channel masks): log-probs 1e-4, loss 1e-4, every parameter gradient within 3e-4 -- what
    the pretrain twin asserts at this depth."""
    from audio8_amd import ops, wav2vec2 as W, ctc as Cc
    from oracle import ref_cpu as R
    ops.set_compute_dtype(torch.float32)
    B, L, V, NL = 4, 16000, 32, 12
    sd = R.init_state_dict(seed=2, num_layers=NL, ctc_labels=V)
    x = synth(B, L, 77)
    lengths = torch.tensor([16000, 13337, 9001, 5120])
    pad_mask = R.sequence_mask(lengths, L)
    T = R.conv_out_lengths(L)[-1]
    tl = torch.tensor([11, 9, 6, 3])
    targets = torch.full((B, 14), 1, dtype=torch.long)
    g = torch.Generator().manual_seed(3)
    for b in range(B):
        targets[b, : tl[b]] = torch.randint(4, V, (int(tl[b]),), generator=g)
        targets[b, tl[b]] = 2
    np.random.seed(6)
    tm = R.create_mask((B, T), 0.5, 10)
    cm = R.create_mask((B, 768), 0.1, 64)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    lp_ref, pm_ref = R.acoustic_model_forward(x, pad_mask, sdg, num_layers=NL, time_mask=tm, channel_mask=cm)
    out_len = pm_ref.sum(-1)
    assert out_len.tolist() == sorted(out_len.tolist(), reverse=True) and int(out_len[-1]) < T // 2      # really ragged
    flat = torch.from_numpy(R.ctc_strip_targets(targets.numpy()))
    loss_ref = F.ctc_loss(lp_ref.transpose(1, 0), flat, out_len, tl, blank=0, reduction="sum", zero_infinity=True)
    loss_ref.backward()
    model = W.create_acoustic_model(V, num_layers=NL, dropout=0.0, freeze_fx=False)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().train()
    model.freeze = False
    np.random.seed(6)
    lp, pm = model(x.cuda(), pad_mask.cuda())
    assert torch.equal(pm.cpu(), pm_ref)
    np.testing.assert_allclose(lp.detach().cpu().numpy(), lp_ref.detach().numpy(), rtol=1e-4, atol=2e-5)
    loss = Cc.CTCLoss(reduction_type="sum")(lp.transpose(1, 0), pm.sum(-1), targets.cuda(), tl)
    loss.backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(loss.item(), loss_ref.item(), rtol=1e-4)
    _grad_check(dict(model.named_parameters()), sdg, 3e-4)


def test_8khz_feature_extractor_parity_f32(restore_dtype):
    """create_model(sample_rate=8): six conv layers (k = 10,3,3,3,2,2; audio8/wav2vec2.py:26-29), total stride 160 -- the fused
    even/odd conv data gradient and the row-remapped GEMMs on a stack other than the 7-layer one."""
    from audio8_amd import ops, wav2vec2 as W
    from oracle import ref_cpu as R
    ops.set_compute_dtype(torch.float32)
    NL, B, L, seed = 2, 2, 12000, 9
    cf = R.CONV_FEATURES[8]
    assert len(cf) == 6 and W.CONV_FEATURES[8] == cf
    T = R.conv_out_lengths(L, cf)[-1]
    assert T == 74
    sd = R.init_state_dict(seed=seed, num_layers=NL, conv_features=cf)
    x = synth(B, L, 21)
    np.random.seed(seed)
    Tm = int(R.create_mask((B, T), 0.65, 10)[0].sum())
    gum = -torch.empty(B * Tm * 2, 320).exponential_(generator=torch.Generator().manual_seed(5)).log()
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    np.random.seed(seed)
    loss_ref, aux = R.pretrain_loss(x, sdg, num_layers=NL, training=True, gumbel=gum, conv_features=cf)
    loss_ref.backward()
    model = W.create_model(sample_rate=8, num_layers=NL, dropout=0.0, dropout_input=0.0, dropout_features=0.0)
    assert len(model.feature_extractor.conv_layers) == 6
    model.load_state_dict(sd, strict=True)
    model = model.cuda().train()
    model.quantizer.gumbel_override = gum.cuda()
    loss_fn = W.create_loss(640, 100)
    np.random.seed(seed)
    loss = loss_fn(model, x.cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert tuple(loss_fn.last_logits.shape) == (B * Tm, 101)
    np.testing.assert_allclose(loss.item(), loss_ref.item(), rtol=1e-4)
    _grad_check(dict(model.named_parameters()), sdg, 1e-4)
    # bf16 mode on the same stack: finite, close to the f32 loss (stated, looser tolerance)
    ops.set_compute_dtype(torch.bfloat16)
    model.zero_grad()
    np.random.seed(seed)
    loss16 = loss_fn(model, x.cuda())
    loss16.backward()
    np.testing.assert_allclose(loss16.item(), loss_ref.item(), rtol=3e-2)


def test_bf16_eval_without_a_stepper_is_deterministic(restore_dtype):
    """Round-6 regression: under torch.no_grad() without a Stepper every encoder layer casts its weights to bf16 on the fly; those casts
    are temporaries, and the half-batch stream still read layer l's while the caching allocator handed their memory to layer l + 1's
    (half of cfg5's frames decoded differently from run to run).  The same eval forward three times, with allocator churn in between,
    must give bit-identical log-probs, close to the f32 mode's."""
    from audio8_amd import ops, wav2vec2 as W
    from oracle import ref_cpu as R
    B, L, V, NL = 6, 48000, 32, 8
    sd = R.init_state_dict(seed=11, num_layers=NL, ctc_labels=V)
    x = synth(B, L, 3).cuda()
    pad = R.sequence_mask(torch.tensor([48000, 41000, 30000, 48000, 20000, 35000]), L).cuda()
    outs = {}
    for dtype in (torch.float32, torch.bfloat16):
        ops.set_compute_dtype(dtype)
        model = W.create_acoustic_model(V, num_layers=NL, dropout=0.0)
        model.load_state_dict(sd, strict=True)
        model = model.cuda().eval()
        model.freeze = False
        runs = []
        for i in range(3):
            with torch.no_grad():
                lp, _ = model(x, pad)
            runs.append(lp.float().clone())
            junk = [torch.empty(int(s), device="cuda") for s in (3e6, 7e5, 1.2e7)[: i + 1]]      # perturb what the allocator hands out next
            del junk
        torch.cuda.synchronize()
        assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2]), dtype
        outs[dtype] = runs[0]
    assert float((outs[torch.float32] - outs[torch.bfloat16]).abs().max()) < 0.15
