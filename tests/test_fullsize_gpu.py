"""Parity at BASELINE.json's full sizes (cfg2: 8 x 250 000 samples, T = 781; cfg5 CTC: T = 624, B = 16) through
size-independent properties, plus direct oracle comparisons wherever the CPU oracle finishes in seconds."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

B, L, T, D, H, DFF = 8, 250000, 781, 768, 12, 3072
M = B * T


def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("layout,m,n,k", [("NT", M, DFF, D), ("NT", M, D, DFF), ("NN", M, D, DFF), ("TN", DFF, D, M)])
def test_gemm_checksum_of_checksums_full_size(layout, m, n, k):
    """C 1 = A (B^T 1): the row sums of the bf16 GEMM output against an f64 matrix-vector product on the host."""
    from audio8_amd import kernels as K, _lib as Lb
    A = rnd((m, k), 1).to(torch.bfloat16)
    Bm = rnd((n, k), 2).to(torch.bfloat16)
    ref = A.double() @ Bm.double().sum(0)                                   # [m]
    dev = "cuda"
    if layout == "NT":
        C = K.gemm(Lb.NT, A.to(dev), Bm.to(dev), m, n, k, out_dtype=torch.float32)
    elif layout == "NN":
        C = K.gemm(Lb.NN, A.to(dev), Bm.T.contiguous().to(dev), m, n, k, out_dtype=torch.float32)
    else:
        C = torch.zeros(m, n, device=dev)
        K.gemm(Lb.TN, A.T.contiguous().to(dev), Bm.T.contiguous().to(dev), m, n, k, out=C, accumulate=True, splitk=3)
    got = C.double().sum(1).cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-4, atol=1e-3 * math.sqrt(k * n))
    # and a sampled set of exact entries
    idx_r = torch.randint(0, m, (64,), generator=torch.Generator().manual_seed(3))
    idx_c = torch.randint(0, n, (64,), generator=torch.Generator().manual_seed(4))
    want = (A[idx_r].double() * Bm[idx_c].double()).sum(1)
    np.testing.assert_allclose(C[idx_r.to(dev), idx_c.to(dev)].double().cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-4 * math.sqrt(k))


def test_conv0_full_length_against_oracle():
    """Conv layer 0 + GroupNorm + GELU at the full 250 000-sample length (the statistics span all 49 999 frames)."""
    from audio8_amd import kernels as K
    x = (0.1 * rnd((B, L), 1)).clamp_(-1, 1)
    w = rnd((512, 1, 10), 2, math.sqrt(2 / 10))
    gw, gb = 1 + 0.1 * rnd((512,), 3), 0.1 * rnd((512,), 4)
    out, stats = K.conv0_gn_gelu_fwd(x.cuda(), w.cuda(), gw.cuda(), gb.cuda(), torch.float32)
    for b in (0, B - 1):                                                   # the oracle on two utterances: ~0.1 s each
        ref = F.gelu(F.group_norm(F.conv1d(x[b:b + 1].unsqueeze(1), w, stride=5), 512, gw, gb, 1e-5)).transpose(1, 2)[0]
        np.testing.assert_allclose(out[b].cpu().numpy(), ref.numpy(), rtol=2e-4, atol=2e-5)
    assert out.shape == (B, 49999, 512)


def test_conv_stack_lengths_and_bf16_vs_f32_full_size():
    from audio8_amd import ops, wav2vec2 as W
    from oracle import ref_cpu as R
    old = ops.compute_dtype()
    try:
        torch.manual_seed(0)
        fx = W.ConvFeatureExtractionModel(W.CONV_FEATURES[16]).cuda()
        x = (0.1 * rnd((B, L), 5)).clamp_(-1, 1).cuda()
        outs = {}
        for dt in (torch.float32, torch.bfloat16):
            ops.set_compute_dtype(dt)
            with torch.no_grad():
                outs[dt] = fx(x).float()
        assert outs[torch.float32].shape == (B, R.conv_out_lengths(L)[-1], 512) == (B, T, 512)
        err = (outs[torch.bfloat16] - outs[torch.float32]).norm() / outs[torch.float32].norm()
        assert err < 3e-2, err
    finally:
        ops.set_compute_dtype(old)


def test_layernorm_rows_are_standardised_full_size():
    from audio8_amd import kernels as K
    x = rnd((M, D), 1, 3.0).cuda() + 5.0
    y, mean, rstd = K.layernorm_fwd(x, None, torch.ones(D, device="cuda"), torch.zeros(D, device="cuda"), 1e-5)
    assert y.mean(1).abs().max().item() < 1e-4
    np.testing.assert_allclose(y.var(1, unbiased=False).cpu().numpy(), 1.0, rtol=1e-3)
    np.testing.assert_allclose(mean.cpu().numpy(), x.mean(1).cpu().numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_partition_of_unity_and_permutation_full_size(dtype):
    """Softmax rows sum to one => with V = ones the output is ones, at T = 781, all 96 (b, h) pairs; and the output
    is invariant to permuting the keys together with their values."""
    from audio8_amd import kernels as K
    d = H * 64
    qkv = rnd((M, 3 * d), 1).to(dtype).cuda()
    qkv[:, 2 * d:] = 1.0
    o, lse = K.attn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, T, H, 3 * d, 0.125)
    np.testing.assert_allclose(o.float().cpu().numpy(), 1.0, rtol=0, atol=2e-6 if dtype == torch.float32 else 8e-3)
    qkv2 = rnd((M, 3 * d), 2).to(dtype).cuda()
    o1, _ = K.attn_fwd(qkv2[:, :d], qkv2[:, d:2 * d], qkv2[:, 2 * d:], B, T, H, 3 * d, 0.125)
    perm = torch.randperm(T, generator=torch.Generator().manual_seed(0)).cuda()
    kv = qkv2.view(B, T, 3 * d)[:, perm].reshape(M, 3 * d)
    mixed = torch.cat([qkv2[:, :d], kv[:, d:]], 1).contiguous()
    o2, _ = K.attn_fwd(mixed[:, :d], mixed[:, d:2 * d], mixed[:, 2 * d:], B, T, H, 3 * d, 0.125)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    np.testing.assert_allclose(o2.float().cpu().numpy(), o1.float().cpu().numpy(), rtol=tol, atol=tol)


def test_encoder_layer_full_size_against_oracle_f32():
    """One post-LN encoder block at the cfg2 size (6248 x 768) in f32 mode against the CPU oracle (~100 GFLOP on the host)."""
    from audio8_amd import ops, layers
    from oracle import ref_cpu as R
    old = ops.compute_dtype()
    try:
        ops.set_compute_dtype(torch.float32)
        sd = {k[len("encoder.transformer.encoders.0."):]: v for k, v in R.init_state_dict(seed=5, num_layers=1).items()
              if k.startswith("encoder.transformer.encoders.0.")}
        layer = layers.TransformerEncoder(H, D, 0.0, d_ff=DFF, layer_norms_after=True)
        layer.load_state_dict(sd)
        layer = layer.cuda().eval()
        x = rnd((B, T, D), 7)
        with torch.no_grad():
            got = layer((x.reshape(M, D).cuda(), None, B, T)).view(B, T, D).cpu()
            ref = R.encoder_layer(x, sd, "", H)
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-4, atol=2e-5)
    finally:
        ops.set_compute_dtype(old)


def test_ctc_full_size_cfg5_against_oracle_and_invariants():
    """cfg5: T = 624 frames, 16 utterances, 32 labels, targets of 50..300 labels.  f64 oracle + the invariant that the
    class gradients of every valid frame sum to zero (posterior occupancies sum to one)."""
    from audio8_amd import kernels as K
    from oracle import ref_cpu as R
    Tn, Bn, V = 624, 16, 32
    g = torch.Generator().manual_seed(11)
    lp = torch.randn(Tn, Bn, V, generator=g).log_softmax(-1)
    tl = torch.randint(50, 301, (Bn,), generator=g)
    il = torch.randint(int(0.6 * Tn), Tn + 1, (Bn,), generator=g)
    il[0], tl[0] = Tn, 300
    il[1], tl[1] = 330, 300                                            # tight but feasible: end-state mass ~1e-40 of the prefix mass
    il[2], tl[2] = 100, 300                                            # infeasible: zero_infinity -> loss 0, gradient 0
    targets = torch.full((Bn, 302), 1, dtype=torch.long)
    for b in range(Bn):
        targets[b, : tl[b]] = torch.randint(4, V, (int(tl[b]),), generator=g)
        targets[b, tl[b]] = 2
    flat = R.ctc_strip_targets(targets.numpy()).astype(np.int32)
    off = np.concatenate([[0], np.cumsum(tl.numpy())]).astype(np.int32)
    nll, grad = K.ctc_fwd_bwd(lp.cuda(), torch.from_numpy(flat).cuda(), torch.from_numpy(off).cuda(), il.int().cuda(), int(tl.max()))
    want_nll, want_grad = R.ctc_loss_np(lp.numpy(), il.numpy(), targets.numpy(), tl.numpy(), reduction="none")
    np.testing.assert_allclose(nll.cpu().numpy(), want_nll, rtol=3e-5, atol=1e-4)
    np.testing.assert_allclose(grad.cpu().numpy(), want_grad, rtol=2e-3, atol=3e-5)
    gs = grad.sum(-1).cpu().numpy()                                       # [T, B]
    feasible = want_nll > 0
    assert feasible[1] and not feasible[2] and nll[2].item() == 0.0 and grad[:, 2].abs().max().item() == 0.0
    for b in range(Bn):
        if feasible[b]:
            assert np.abs(gs[: int(il[b]), b]).max() < 5e-5
        assert np.abs(grad[int(il[b]):, b].cpu().numpy()).max() == 0 if il[b] < Tn else True


def test_full_pretrain_step_cfg2_masks_negatives_and_finiteness():
    """The whole cfg2 step in bf16: host index sampling bit-exact against the oracle for the same seed, equal mask
    count per row, finite loss close to the f32 mode's, non-zero gradients everywhere."""
    from audio8_amd import ops, wav2vec2 as W
    from oracle import ref_cpu as R
    old = ops.compute_dtype()
    try:
        torch.manual_seed(0)
        model = W.create_model(num_layers=12, dropout=0.0, dropout_input=0.0, dropout_features=0.0).cuda().train()
        x = (0.1 * rnd((B, L), 1234)).clamp_(-1, 1).cuda()
        loss_fn = W.create_loss(640, 100)
        vals = {}
        for dt in (torch.bfloat16, torch.float32):
            ops.set_compute_dtype(dt)
            model.zero_grad()
            np.random.seed(0)
            torch.manual_seed(1)
            ctx, q, ppl, tm = model(x)
            np.random.seed(0)
            want_mask = R.create_mask((B, T), 0.65, 10)
            assert np.array_equal(tm.cpu().numpy(), want_mask) and len(set(want_mask.sum(1))) == 1
            np.random.seed(0)
            torch.manual_seed(1)
            loss = loss_fn(model, x)
            loss.backward()
            vals[dt] = loss.item()
            assert math.isfinite(vals[dt])
            for n, p in model.named_parameters():
                assert p.grad is not None and torch.isfinite(p.grad).all(), n
            assert model.feature_extractor.conv_layers[0][0].weight.grad.abs().sum() > 0
        assert abs(vals[torch.bfloat16] - vals[torch.float32]) < 3e-2 * abs(vals[torch.float32])
    finally:
        ops.set_compute_dtype(old)


def test_full_pretrain_step_cfg4_large_model():
    """cfg4: wav2vec2-LARGE (24 layers, d_model 1024, 16 heads, d_ff 4096), 4 x 320 000 samples (T = 999).  Index sampling
    bit-exact against the oracle, equal mask count per row, bf16 loss close to the f32 mode's, finite non-zero gradients on
    every one of the 316 M parameters."""
    from audio8_amd import ops, wav2vec2 as W
    from oracle import ref_cpu as R
    Bc, Lc, Tc = 4, 320000, 999
    old = ops.compute_dtype()
    try:
        torch.manual_seed(0)
        model = W.create_model(d_model=1024, num_heads=16, num_layers=24, d_ff=4096, dropout=0.0, dropout_input=0.0,
                               dropout_features=0.0).cuda().train()
        assert sum(p.numel() for p in model.parameters()) > 3.1e8
        x = (0.1 * rnd((Bc, Lc), 4321)).clamp_(-1, 1).cuda()
        loss_fn = W.create_loss(640, 100)
        vals = {}
        for dt in (torch.bfloat16, torch.float32):
            ops.set_compute_dtype(dt)
            model.zero_grad()
            np.random.seed(0)
            torch.manual_seed(1)
            ctx, q, ppl, tm = model(x)
            assert ctx.shape == (Bc, Tc, 256) and q.shape[0] == Bc and q.shape[2] == 256
            np.random.seed(0)
            want_mask = R.create_mask((Bc, Tc), 0.65, 10)
            assert np.array_equal(tm.cpu().numpy(), want_mask) and len(set(want_mask.sum(1))) == 1 and q.shape[1] == want_mask[0].sum()
            np.random.seed(0)
            torch.manual_seed(1)
            loss = loss_fn(model, x)
            loss.backward()
            vals[dt] = loss.item()
            assert math.isfinite(vals[dt])
            lg = loss_fn.last_logits
            assert lg.shape == (Bc * int(want_mask[0].sum()), 101) and float(lg.abs().max()) <= 1.0 + 1e-3        # cosine similarities
            for n, p in model.named_parameters():
                assert p.grad is not None and torch.isfinite(p.grad).all(), n
            assert model.encoder.transformer.encoders[23].ffn[0].layer.weight.grad.abs().sum() > 0
            assert model.feature_extractor.conv_layers[0][0].weight.grad.abs().sum() > 0
        assert abs(vals[torch.bfloat16] - vals[torch.float32]) < 3e-2 * abs(vals[torch.float32])
    finally:
        ops.set_compute_dtype(old)


def test_full_ctc_finetune_step_cfg5():
    """cfg5: the train.py step at full size -- wav2vec2-base + 32-label head, 16 x 200 000 samples with ragged lengths
    (uniform in [0.6 L, L]), freeze_fx=True, encoder unfrozen, training-time time/channel masks.  Checks: frame-level pad mask
    equals the reference's down-sampling rule; every frame's log-probs are normalised; the CTC loss of the model's own
    log-probs equals the f64 oracle's; gradients reach every encoder parameter and none of the frozen feature extractor."""
    from audio8_amd import ops, wav2vec2 as W, ctc as Cc
    from oracle import ref_cpu as R
    Bc, Lc, Tc, V = 16, 200000, 624, 32
    old = ops.compute_dtype()
    try:
        ops.set_compute_dtype(torch.bfloat16)
        torch.manual_seed(0)
        model = W.create_acoustic_model(V).cuda().train()
        model.freeze = False
        g = torch.Generator().manual_seed(1234)
        x = (0.1 * rnd((Bc, Lc), 99)).clamp_(-1, 1)
        lengths = torch.randint(int(0.6 * Lc), Lc + 1, (Bc,), generator=g)
        lengths[0] = Lc
        pad = R.sequence_mask(lengths, Lc)
        tl = torch.randint(50, 301, (Bc,), generator=g)
        targets = torch.full((Bc, 302), 1, dtype=torch.long)
        for b in range(Bc):
            targets[b, : tl[b]] = torch.randint(4, V, (int(tl[b]),), generator=g)
            targets[b, tl[b]] = 2
        np.random.seed(3)
        lp, pm = model(x.cuda(), pad.cuda())
        assert lp.shape == (Bc, Tc, V) and lp.dtype == torch.float32
        want_pm = R.downsample_pad_mask(pad, Tc)
        assert torch.equal(pm.cpu(), want_pm) and int(want_pm[0].sum()) == Tc and int(want_pm.sum(-1).min()) < Tc
        np.testing.assert_allclose(lp.detach().exp().sum(-1).cpu().numpy(), 1.0, rtol=2e-5)
        out_len = pm.sum(-1)
        loss = Cc.CTCLoss(reduction_type="sum")(lp.transpose(1, 0), out_len, targets.cuda(), tl)
        loss.backward()
        want_nll, _ = R.ctc_loss_np(lp.detach().transpose(1, 0).cpu().numpy(), out_len.cpu().numpy(), targets.numpy(), tl.numpy(), reduction="none")
        np.testing.assert_allclose(loss.item(), float(want_nll.sum()), rtol=1e-4)
        assert math.isfinite(loss.item()) and loss.item() > 0
        for n, p in model.named_parameters():
            if n.startswith("encoder.feature_extractor."):
                assert p.grad is None, n                                  # freeze_fx=True (audio8/wav2vec2.py:697-698)
            else:
                assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().sum() > 0, n
    finally:
        ops.set_compute_dtype(old)


def test_cfg2_sequence_length_bf16_against_oracle():
    """The benchmark dtype at the benchmark depth AND sequence length against the oracle: wav2vec2-base (12 layers), 2 x 250 000 samples
    (T = 781 frames, 64-key attention tiles with a ragged last tile, conv layer 1 at 25 k rows per utterance); the oracle runs this in a
    few seconds on the host.  Stated bf16 tolerance: loss within 2e-3, the whole gradient within 12 % (relative L2) of the oracle's and
    cosine > 0.995 (a wrong constant in a bf16-only kernel is caught at kernel level: test_attention_bf16_tolerance_detects_a_wrong_scale); per-parameter detail is test_model_gpu.py::test_pretrain_step_parity_bf16_full_depth_base."""
    from test_model_gpu import _pretrain_pair
    from audio8_amd import ops
    old = ops.compute_dtype()
    try:
        model, loss, loss_fn, sdg, loss_ref, aux = _pretrain_pair(torch.bfloat16, num_layers=12, B=2, L=250000, seed=1)
        np.testing.assert_allclose(loss.item(), loss_ref.item(), rtol=2e-3)
        got = {k: p.grad.detach().float().cpu() for k, p in model.named_parameters()}
        allg = torch.cat([got[k].flatten() for k in sdg])
        allr = torch.cat([sdg[k].grad.flatten() for k in sdg])
        gcos = float(torch.nn.functional.cosine_similarity(allg, allr, dim=0))
        grel = float((allg - allr).norm() / allr.norm())
        print("bf16, 12 layers, T = 781: loss %.5f vs %.5f, global cosine %.4f, global relative error %.3f" % (loss.item(), loss_ref.item(), gcos, grel))
        assert gcos > 0.995 and grel < 0.12, (gcos, grel)          # measured 0.9982 / 0.069 (rounds 3-5); the band was 0.98 / 0.25 until round 5
    finally:
        ops.set_compute_dtype(old)


def test_cfg2_sequence_length_f32_against_oracle():
    """north_star's tolerance mode (f32) at the benchmark depth AND sequence length, with every gradient: wav2vec2-base (12 layers),
    2 x 250 000 samples -> T = 781 frames (the 64-key ragged last attention tile, conv layer 1 at 25 k rows per utterance, the
    Tm = 368 loss geometry).  Loss and contrastive logits within 1e-4 rtol (audio8/wav2vec2.py:927-952, 371-392), every parameter
    gradient within 3e-4 relative L2 of the oracle's (the 12-layer bound of test_model_gpu.py)."""
    from test_model_gpu import _pretrain_pair, _check_f32_parity
    from audio8_amd import ops
    old = ops.compute_dtype()
    try:
        worst = _check_f32_parity(*_pretrain_pair(torch.float32, num_layers=12, B=2, L=250000, seed=1), grad_tol=3e-4)
        print("f32, 12 layers, T = 781: worst per-parameter relative gradient error %.3e" % worst)
    finally:
        ops.set_compute_dtype(old)
