"""The CTC fine-tune head for any vocabulary size (audio8/train.py:194-198 builds `num_labels = len(vocab)` for letter or
BPE vocabularies), the label-strided log-softmax kernels, device-side target packing, and the optimizer's treatment of
parameters that got no gradient (frozen encoder of audio8/train.py:296-297)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def synth(B, L, seed=1234):
    g = torch.Generator().manual_seed(seed)
    return (0.1 * torch.randn(B, L, generator=g)).clamp_(-1, 1)


def rel_err(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-7 * (b.numel() ** 0.5)))


@pytest.fixture
def restore_dtype():
    from audio8_amd import ops
    old = ops.compute_dtype()
    yield
    ops.set_compute_dtype(old)


@pytest.mark.parametrize("cols,ldx", [(1, 8), (31, 32), (32, 32), (64, 64), (65, 72), (1000, 1000), (1003, 1008)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_log_softmax_any_width(cols, ldx, dtype):
    from audio8_amd import kernels as K
    g = torch.Generator().manual_seed(cols)
    rows = 257
    x = (3 * torch.randn(rows, ldx, generator=g)).to(dtype)
    x[:, cols:] = 77.0                                  # padding columns must be ignored
    y = K.log_softmax_fwd(x.cuda(), cols)
    want = F.log_softmax(x[:, :cols].float(), -1)
    np.testing.assert_allclose(y.cpu().numpy(), want.numpy(), rtol=1e-5, atol=2e-6)
    dy = torch.randn(rows, cols, generator=g)
    dx = K.log_softmax_bwd(y, dy.cuda(), dtype, ldx)
    xr = x[:, :cols].float().requires_grad_(True)
    (F.log_softmax(xr, -1) * dy).sum().backward()
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    np.testing.assert_allclose(dx[:, :cols].float().cpu().numpy(), xr.grad.numpy(), rtol=tol, atol=tol)
    assert dx.shape == (rows, ldx) and float(dx[:, cols:].float().abs().sum()) == 0.0


@pytest.mark.parametrize("V", [31, 32, 1000])
def test_acoustic_head_any_vocabulary_f32(restore_dtype, V):
    """Wav2Vec2AcousticModel with V labels (not a multiple of 8 / far beyond one wave): log-probs, CTC loss and every gradient
    against the oracle."""
    from audio8_amd import ops, wav2vec2 as W, ctc as Cc
    from oracle import ref_cpu as R
    ops.set_compute_dtype(torch.float32)
    B, L = 2, 16000
    sd = R.init_state_dict(seed=V, num_layers=1, ctc_labels=V)
    x = synth(B, L, 5)
    lengths = torch.tensor([16000, 11000])
    pad_mask = R.sequence_mask(lengths, L)
    tl = torch.tensor([9, 6])
    targets = torch.full((B, 12), 1, dtype=torch.long)
    g = torch.Generator().manual_seed(3)
    for b in range(B):
        targets[b, : tl[b]] = torch.randint(4, V, (int(tl[b]),), generator=g)
        targets[b, tl[b]] = 2
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    lp_ref, pm_ref = R.acoustic_model_forward(x, pad_mask, sdg, num_layers=1)
    flat = torch.from_numpy(R.ctc_strip_targets(targets.numpy()))
    loss_ref = F.ctc_loss(lp_ref.transpose(1, 0), flat, pm_ref.sum(-1), tl, blank=0, reduction="sum", zero_infinity=True)
    loss_ref.backward()
    model = W.create_acoustic_model(V, num_layers=1, dropout=0.0, timestep_masking=0.0, channel_masking=0.0, freeze_fx=False)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().train()
    model.freeze = False
    lp, pm = model(x.cuda(), pad_mask.cuda())
    assert lp.shape == (B, pm.shape[1], V) and lp.dtype == torch.float32
    np.testing.assert_allclose(lp.detach().cpu().numpy(), lp_ref.detach().numpy(), rtol=1e-4, atol=2e-5)
    loss = Cc.CTCLoss(reduction_type="sum")(lp.transpose(1, 0), pm.sum(-1), targets.cuda(), tl)
    loss.backward()
    np.testing.assert_allclose(loss.item(), loss_ref.item(), rtol=1e-4)
    got = dict(model.named_parameters())
    for k, ref in sdg.items():
        if k.endswith("w_K.layer.bias"):
            continue
        if ref.grad is None:                                   # mask_emb: no time masking in this test
            assert got[k].grad is None, k
            continue
        assert got[k].grad is not None and got[k].grad.shape == ref.grad.shape, k
        e = rel_err(got[k].grad, ref.grad)
        assert e < 2e-3, f"{k}: rel grad err {e}"
    # bf16 throughput mode runs the same head
    ops.set_compute_dtype(torch.bfloat16)
    with torch.no_grad():
        lp16, _ = model(x.cuda(), pad_mask.cuda())
    assert lp16.shape == lp.shape and torch.isfinite(lp16).all()
    np.testing.assert_allclose(lp16.exp().sum(-1).cpu().numpy(), 1.0, rtol=1e-3)


def test_ctc_loss_device_targets_match_host_targets():
    """targets on the device (audio8/train.py:35): PAD/EOS stripping on the GPU gives the same loss and gradient as the host
    path, and both equal F.ctc_loss; `mean` reduction included."""
    from audio8_amd import ctc as Cc
    Tn, Bn, V = 90, 5, 32
    g = torch.Generator().manual_seed(1)
    lp = torch.randn(Tn, Bn, V, generator=g).log_softmax(-1)
    tl = torch.tensor([20, 1, 0, 33, 7])
    il = torch.tensor([90, 60, 50, 80, 90])
    targets = torch.full((Bn, 40), 1, dtype=torch.long)
    for b in range(Bn):
        targets[b, : tl[b]] = torch.randint(4, V, (int(tl[b]),), generator=g)
        targets[b, tl[b]] = 2                              # EOS, then PAD
    flat = targets[(targets != 1) & (targets != 2)]
    for red in ("sum", "mean"):
        res = []
        for tg in (targets.cuda(), targets):
            x = lp.clone().cuda().requires_grad_(True)
            loss = Cc.CTCLoss(reduction_type=red)(x, il.cuda(), tg, tl)
            loss.backward()
            res.append((loss.item(), x.grad.cpu()))
        xr = lp.clone().requires_grad_(True)
        want = F.ctc_loss(xr, flat, il, tl, blank=0, reduction=red, zero_infinity=True)
        want.backward()
        assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])
        np.testing.assert_allclose(res[0][0], want.item(), rtol=1e-5)
        np.testing.assert_allclose(res[0][1].numpy(), xr.grad.numpy(), rtol=2e-3, atol=2e-5)


def test_ctc_loss_device_targets_wide_padding_and_deferred_length_check():
    """Device targets padded far wider than any real label sequence (width 700 > the kernel's 511-label bound): the bound comes from
    target_lengths, not from the padding.  A row whose stated length disagrees with its non-PAD/EOS count must not shift the other
    rows' labels (each row is packed at its own offset) and is reported at the next call."""
    from audio8_amd import ctc as Cc
    Tn, Bn, V = 120, 4, 32
    g = torch.Generator().manual_seed(2)
    lp = torch.randn(Tn, Bn, V, generator=g).log_softmax(-1)
    tl = torch.tensor([30, 12, 45, 5])
    il = torch.tensor([120, 100, 120, 90])
    targets = torch.full((Bn, 700), 1, dtype=torch.long)
    for b in range(Bn):
        targets[b, : tl[b]] = torch.randint(4, V, (int(tl[b]),), generator=g)
        targets[b, tl[b]] = 2
    flat = targets[(targets != 1) & (targets != 2)]
    loss_fn = Cc.CTCLoss(reduction_type="sum")
    for tlen in (tl, tl.cuda()):                            # host lengths: exact bound; device lengths: one sync because 700 > 511
        x = lp.clone().cuda().requires_grad_(True)
        loss = loss_fn(x, il.cuda(), targets.cuda(), tlen)
        loss.backward()
        xr = lp.clone().requires_grad_(True)
        want = F.ctc_loss(xr, flat, il, tl, blank=0, reduction="sum", zero_infinity=True)
        want.backward()
        np.testing.assert_allclose(loss.item(), want.item(), rtol=1e-5)
        np.testing.assert_allclose(x.grad.cpu().numpy(), xr.grad.numpy(), rtol=2e-3, atol=2e-5)
    # row 1 claims 10 labels but holds 12: rows 0, 2, 3 keep their exact per-utterance losses; the mismatch surfaces one call later
    bad_tl = tl.clone()
    bad_tl[1] = 10
    per = Cc.CTCLoss(reduction_type="none")
    nll_bad = per(lp.cuda(), il.cuda(), targets.cuda(), bad_tl).cpu()
    nll_ok = F.ctc_loss(lp, flat, il, tl, blank=0, reduction="none", zero_infinity=True)
    np.testing.assert_allclose(nll_bad[[0, 2, 3]].numpy(), nll_ok[[0, 2, 3]].numpy(), rtol=1e-5)
    with pytest.raises(RuntimeError, match="did not match"):
        per(lp.cuda(), il.cuda(), targets.cuda(), tl)
    assert torch.isfinite(per(lp.cuda(), il.cuda(), targets.cuda(), tl)).all()          # the flag was consumed; a clean batch runs


def test_stepper_leaves_gradless_parameters_alone(restore_dtype):
    """train.py starts with the encoder frozen (`model.freeze = True`): only `proj` receives gradients.  torch's AdamW skips
    `p.grad is None`; the fused update must too -- no weight decay, no moment decay, own step count -- and a parameter
    that starts training later begins at step 1 (bias correction restarts)."""
    from audio8_amd import ops, wav2vec2 as W, ctc as Cc
    from audio8_amd.trainer import Stepper
    ops.set_compute_dtype(torch.float32)
    V = 32
    torch.manual_seed(0)
    model = W.create_acoustic_model(V, num_layers=1, dropout=0.0, timestep_masking=0.0, channel_masking=0.0).cuda().train()
    ref = W.create_acoustic_model(V, num_layers=1, dropout=0.0, timestep_masking=0.0, channel_masking=0.0).cuda().train()
    ref.load_state_dict(model.state_dict())
    x = synth(2, 16000, 3).cuda()
    pm = torch.ones(2, 16000, dtype=torch.bool, device="cuda")
    targets = torch.tensor([[5, 6, 7, 2, 1], [8, 9, 2, 1, 1]], device="cuda")
    tl = torch.tensor([3, 2])
    loss_fn = Cc.CTCLoss()

    def run(m):
        lp, pmo = m(x, pm)
        return loss_fn(lp.transpose(1, 0), pmo.sum(-1), targets, tl)

    st = Stepper(model, lr=1e-2, weight_decay=0.1, clip=0.0)
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-2, weight_decay=0.1)
    try:
        for step in range(3):
            if step == 2:
                model.freeze = ref.freeze = False          # `unfreeze_enc_after_step` reached (freeze_fx stays True)
            st.zero_grad()
            run(model).backward()
            st.step()
            opt.zero_grad(set_to_none=True)
            run(ref).backward()
            opt.step()
            torch.cuda.synchronize()
            for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
                if k.endswith("w_K.layer.bias"):               # analytically zero gradient: Adam turns its rounding noise into +-lr steps
                    continue
                if q.grad is None:
                    assert torch.equal(p.detach(), q.detach()), (step, k)          # bit-identical: never touched
                else:       # elements whose gradient is rounding noise around zero take a +-lr step of either sign under Adam: allow a handful
                    a, bq = p.detach().cpu().numpy(), q.detach().cpu().numpy()
                    bad = np.abs(a - bq) > 2e-4 + 1e-4 * np.abs(bq)
                    assert bad.mean() < 1e-4 and np.abs(a - bq).max() <= 2.1e-2 * (step + 1), (step, k, int(bad.sum()))
        pos = {id(q): i for i, q in enumerate(st.state.params)}                 # FlatState order (q/k/v made adjacent), not module order
        named = dict(model.named_parameters())
        steps = {n: st.param_steps[pos[id(q)]] for n, q in named.items()}
        assert steps["proj.weight"] == 3 and steps["encoder.encoder.ln.weight"] == 1
        assert steps["encoder.feature_extractor.conv_layers.1.0.weight"] == 0
        o = st.state.offsets[pos[id(named["encoder.feature_extractor.conv_layers.1.0.weight"])]]
        assert float(st.state.flat_m[o:o + 100].abs().sum()) == 0.0
        # a caller that drops the flat gradient views (model.zero_grad(set_to_none=True) instead of stepper.zero_grad()) is told so
        model.zero_grad(set_to_none=True)
        run(model).backward()
        with pytest.raises(RuntimeError, match="flat gradient buffer"):
            st.step()
    finally:
        st.close()


def test_prefetched_plan_for_another_shape_is_undone(restore_dtype):
    """INTEGRATION: the reference loader yields a different [B, L] almost every step.  A plan prefetched for the wrong shape
    must not raise and must not disturb the numpy RNG order: the forward falls back to inline draws with indices
    bit-identical to a run that never prefetched."""
    from audio8_amd import ops, wav2vec2 as W
    ops.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(0)
    model = W.create_model(num_layers=1, dropout=0.0, dropout_input=0.0, dropout_features=0.0).cuda().train()
    loss_fn = W.create_loss(640, 100)
    xa, xb = synth(2, 16000, 1).cuda(), synth(3, 12000, 2).cuda()

    def run(prefetch_wrong):
        np.random.seed(5)
        out = []
        loss_fn(model, xa)
        out.append(loss_fn.last_logits.shape)
        if prefetch_wrong:
            W.prefetch_host_draws(model, loss_fn, xa.shape)          # the NEXT batch turns out to be xb
        _, _, _, tm = model(xb)
        out.append(tm.cpu().numpy().copy())
        _, _, _, tm2 = model(xa)
        out.append(tm2.cpu().numpy().copy())
        out.append(np.random.random())
        return out
    a, b = run(False), run(True)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3]


@pytest.mark.parametrize("rpg", [1, 2, 3, 5, 8, 21])
def test_contrastive_backward_small_groups(rpg):
    """rows_per_group below 8 (very short clips: Tm = 1..7) must not overrun the scratch buffer: every group size gives the
    gradients of the ungrouped path."""
    from audio8_amd import kernels as K
    g = torch.Generator().manual_seed(rpg)
    B, N, C = 4, 100, 256
    rows = B * rpg
    ctx, lat = torch.randn(rows, C, generator=g).cuda(), torch.randn(rows, C, generator=g).cuda()
    idx = torch.stack([torch.randint(0, rpg, (rpg * N,), generator=g) + b * rpg for b in range(B)]).view(rows, N).int().cuda()
    logits, loss_sum, inv = K.contrastive_fwd(ctx, lat, idx)
    gs = torch.full((1,), 1.0 / rows, device="cuda")
    guard = torch.full((1 << 20,), 7.0, device="cuda")              # neighbours of the scratch allocation
    d0 = K.contrastive_bwd(ctx, lat, idx, logits, gs, 0, inv)
    d1 = K.contrastive_bwd(ctx, lat, idx, logits, gs, rpg, inv)
    torch.cuda.synchronize()
    assert float(guard.min()) == 7.0 and float(guard.max()) == 7.0
    for a, b in zip(d0, d1):
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=2e-4, atol=1e-6)
