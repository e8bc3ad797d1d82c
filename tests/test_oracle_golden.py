"""Oracle (oracle/ref_cpu.py) pinned against golden vectors produced by the reference's own
code (tools/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from tools.gen_golden import conv_fixture_weights


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_g1_create_mask_bit_exact(golden_dir):
    g = _load(golden_dir, "g1_masks.npz")
    for key in g["keys"]:
        key = str(key)
        seed = int(key.split("_")[0][1:])
        B, T = (int(v) for v in g[key + "_shape"])
        p = int(key.split("_p")[1].split("_")[0]) / 100.0
        ml = int(key.split("_l")[1])
        np.random.seed(seed)
        m = R.create_mask((B, T), p, ml)
        tail = np.random.rand()
        want = np.unpackbits(g[key])[: B * T].reshape(B, T).astype(bool)
        assert np.array_equal(m, want), key
        assert tail == float(g[key + "_tail"]), f"RNG draw count differs: {key}"
        assert len(set(m.sum(1).tolist())) == 1  # equal count per row


def test_g2_negatives_bit_exact(golden_dir):
    g = _load(golden_dir, "g2_negatives.npz")
    for key in g["keys"]:
        key = str(key)
        seed = int(key.split("_")[0][1:])
        B = int(key.split("_b")[1].split("_")[0])
        Tm = int(key.split("_t")[1].split("_")[0])
        N = int(key.split("_n")[1])
        np.random.seed(seed)
        idx = R.sample_negative_indices(B, Tm, N)
        assert np.array_equal(idx, g[key].astype(np.int64)), key
        own = np.repeat(np.arange(Tm), N)[None, :] + (np.arange(B) * Tm)[:, None]
        assert not (idx == own).any()
    np.random.seed(7)
    m = R.create_mask((2, 49), 0.65, 10)
    idx = R.sample_negative_indices(2, int(m[0].sum()), 100)
    assert np.array_equal(m, g["order_mask"])
    assert np.array_equal(idx, g["order_idx"].astype(np.int64))


def _conv_sd(seed):
    ws, gn_w, gn_b = conv_fixture_weights(seed)
    sd = {f"feature_extractor.conv_layers.{i}.0.weight": torch.from_numpy(w).requires_grad_(True) for i, w in enumerate(ws)}
    sd["feature_extractor.conv_layers.0.2.weight"] = torch.from_numpy(gn_w).requires_grad_(True)
    sd["feature_extractor.conv_layers.0.2.bias"] = torch.from_numpy(gn_b).requires_grad_(True)
    return sd


@pytest.mark.parametrize("tag", ["a", "b"])
def test_g3_conv_stack(golden_dir, tag):
    g = _load(golden_dir, "g3_conv.npz")
    sd = _conv_sd(int(g[f"{tag}_seed"]))
    x = torch.from_numpy(g[f"{tag}_x"]).requires_grad_(True)
    outs = R.conv_feature_extractor(x, sd, return_all=True)
    y = outs[-1]
    np.testing.assert_allclose(y.detach().numpy(), g[f"{tag}_y"], rtol=1e-5, atol=1e-6)
    for o, st in zip(outs, g[f"{tag}_layer_stats"]):
        assert o.shape[-1] == int(st[2])
        np.testing.assert_allclose(o.double().mean().item(), st[0], rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(o.double().abs().mean().item(), st[1], rtol=1e-4)
    assert [o.shape[-1] for o in outs] == R.conv_out_lengths(x.shape[1])
    y.backward(torch.from_numpy(g[f"{tag}_gy"]))
    np.testing.assert_allclose(x.grad.numpy(), g[f"{tag}_gx"], rtol=1e-4, atol=1e-6)
    for i in range(7):
        gw = sd[f"feature_extractor.conv_layers.{i}.0.weight"].grad.numpy().reshape(-1)
        sub = gw[::97] if gw.size > 100000 else gw
        np.testing.assert_allclose(sub, g[f"{tag}_gw{i}_sub"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(np.sqrt((gw.astype(np.float64) ** 2).sum()), g[f"{tag}_gw{i}_norm"], rtol=1e-5)
    np.testing.assert_allclose(sd["feature_extractor.conv_layers.0.2.weight"].grad.numpy(), g[f"{tag}_g_gn_w"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(sd["feature_extractor.conv_layers.0.2.bias"].grad.numpy(), g[f"{tag}_g_gn_b"], rtol=1e-4, atol=1e-5)


def test_g4_vq(golden_dir):
    g = _load(golden_dir, "g4_vq.npz")
    assert float(g["tau"]) == R.quantizer_temperature() == 0.5
    assert float(g["tau_after_set_num_updates_1000"]) == R.quantizer_temperature(1000) == 2
    sd = {"quantizer.vars": torch.from_numpy(g["vars"]).requires_grad_(True),
          "quantizer.weight_proj.weight": torch.from_numpy(g["wp"]).requires_grad_(True),
          "quantizer.weight_proj.bias": torch.from_numpy(g["bp"]).requires_grad_(True)}
    y = torch.from_numpy(g["y"])
    q, ppl = R.gumbel_vq(y, sd, training=False)
    np.testing.assert_allclose(q.detach().numpy(), g["eval_q"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(ppl.item(), g["eval_ppl"], rtol=1e-6)
    yy = y.clone().requires_grad_(True)
    q, ppl = R.gumbel_vq(yy, sd, tau=0.5, training=True, gumbel=torch.from_numpy(g["gumbel"]))
    np.testing.assert_allclose(q.detach().numpy(), g["train_q"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ppl.item(), g["train_ppl"], rtol=1e-6)
    (q * torch.from_numpy(g["train_gq"])).sum().backward(retain_graph=True)
    np.testing.assert_allclose(yy.grad.numpy(), g["train_gy"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(sd["quantizer.vars"].grad.numpy(), g["train_gvars"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(sd["quantizer.weight_proj.weight"].grad.numpy(), g["train_gwp"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(sd["quantizer.weight_proj.bias"].grad.numpy(), g["train_gbp"], rtol=1e-4, atol=1e-5)
    yy2 = y.clone().requires_grad_(True)
    _, ppl2 = R.gumbel_vq(yy2, sd, tau=0.5, training=True, gumbel=torch.from_numpy(g["gumbel"]))
    ppl2.backward()
    np.testing.assert_allclose(yy2.grad.numpy(), g["ppl_gy"], rtol=1e-4, atol=1e-6)


def test_g5_loss(golden_dir):
    g = _load(golden_dir, "g5_loss.npz")
    outputs = torch.from_numpy(g["outputs"]).requires_grad_(True)
    latents = torch.from_numpy(g["latents"]).requires_grad_(True)
    gs = torch.tensor(float(g["gs"]), requires_grad=True)
    mask = torch.from_numpy(g["mask"])
    idx = torch.from_numpy(g["idx"].astype(np.int64))
    np.random.seed(9)
    mine = R.sample_negative_indices(2, 21, 100)
    assert np.array_equal(mine, idx.numpy())
    B, T, C = outputs.shape
    om = outputs[mask.unsqueeze(-1).expand_as(outputs)].view(B, -1, C)
    logits = R.contrastive_logits(om, latents, idx)
    np.testing.assert_allclose(logits.detach().numpy(), g["logits"], rtol=1e-6, atol=1e-7)
    loss = R.wav2vec2_loss(outputs, latents, gs, mask, idx, int(g["n_vars"]))
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-6)
    loss.backward()
    np.testing.assert_allclose(outputs.grad.numpy(), g["g_outputs"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(latents.grad.numpy(), g["g_latents"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(gs.grad.item(), g["g_gs"], rtol=1e-6)


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("red", ["sum", "mean"])
def test_g6_ctc(golden_dir, tag, red):
    g = _load(golden_dir, "g6_ctc.npz")
    loss, grad = R.ctc_loss_np(g[f"{tag}_lp"], g[f"{tag}_il"], g[f"{tag}_targets"], g[f"{tag}_tl"], reduction=red)
    np.testing.assert_allclose(loss, g[f"{tag}_{red}_loss"], rtol=1e-5)
    # The golden gradient was produced by ATen's fp32 CPU kernel: at T=624 its log-alpha/beta
    # values reach ~-2000 where one fp32 ulp is 1.2e-4, so the golden itself carries ~1e-3 of
    # absolute noise (measured 1.3e-3 max vs float64).  The float64 cross-check below is exact.
    atol = 1e-6 if tag == "a" else 2.5e-3
    np.testing.assert_allclose(grad, g[f"{tag}_{red}_grad"], rtol=1e-4, atol=atol)
    lp = torch.from_numpy(g[f"{tag}_lp"]).double().requires_grad_(True)
    flat = torch.from_numpy(R.ctc_strip_targets(g[f"{tag}_targets"]))
    l64 = torch.nn.functional.ctc_loss(lp, flat, torch.from_numpy(g[f"{tag}_il"]), torch.from_numpy(g[f"{tag}_tl"]),
                                       blank=0, reduction=red, zero_infinity=True)
    l64.backward()
    np.testing.assert_allclose(loss, l64.item(), rtol=1e-10)
    np.testing.assert_allclose(grad, lp.grad.numpy(), rtol=1e-7, atol=1e-10)


def test_ctc_metrics_greedy():
    vocab = {0: '<s>', 1: '<pad>', 2: '</s>', 3: '<unk>', 4: '|', 5: 'a', 6: 'b', 7: 'c'}
    # frames: a a <blank> a | b b c  -> "aa bc"
    seq = [5, 5, 0, 5, 4, 6, 6, 7]
    lp = torch.full((1, len(seq), 8), -10.0)
    for t, s in enumerate(seq):
        lp[0, t, s] = 0.0
    tgt = torch.tensor([[5, 5, 4, 6, 6, 2, 1, 1]])   # "aa bb"
    m = R.ctc_metrics(lp, tgt, [10 ** 6], vocab)
    assert m == dict(wv_errors=1, w_errors=1, w_total=2, c_errors=1, c_total=5)
    assert R.edit_distance("kitten", "sitting") == 3


def test_model_forward_shapes_and_keys():
    """cfg1-shaped plumbing: restated model runs end to end; key names match the reference maps."""
    sd = R.init_state_dict(seed=0, num_layers=2)
    np.random.seed(0)
    torch.manual_seed(0)
    x = (0.1 * torch.randn(2, 16000)).clamp_(-1, 1)
    loss, aux = R.pretrain_loss(x, sd, num_layers=2, training=False)
    assert aux["ctx"].shape == (2, 49, 256) and aux["q"].shape == (2, 21, 256)
    assert aux["time_mask"].sum(1).tolist() == [21, 21]
    assert torch.isfinite(loss)
    assert "encoder.transformer.encoders.1.ffn.3.layer.weight" in sd and "encoder.pos_conv.conv.1.weight_g" in sd
    sdc = R.init_state_dict(seed=0, num_layers=1, ctc_labels=32)
    pm = R.sequence_mask(torch.tensor([16000, 9000]), 16000)
    lp, pm2 = R.acoustic_model_forward(x, pm, sdc, num_layers=1)
    assert lp.shape == (2, 49, 32) and pm2.shape == (2, 49) and pm2[1].sum() == 9000 // (16000 // 49)
