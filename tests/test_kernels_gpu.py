"""GPU parity of every HIP kernel (through the C ABI) against the CPU oracle / golden vectors."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-2}


def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def close(got, ref, tol, what=""):
    got = got.detach().float().cpu().numpy()
    ref = ref.detach().float().cpu().numpy()
    scale = max(1.0, float(np.abs(ref).max()))
    np.testing.assert_allclose(got, ref, rtol=tol, atol=tol * scale, err_msg=what)


# ------------------------------------------------------------------ conv layer 0
@pytest.mark.parametrize("dtype", DT)
def test_conv0_gn_gelu(dtype):
    from audio8_amd import kernels as K
    B, L, C = 2, 4000, 512
    x = rnd((B, L), 1, 0.1).clamp_(-1, 1)
    w = rnd((C, 1, 10), 2, math.sqrt(2 / 10)).requires_grad_(True)
    gw = (1 + 0.1 * rnd((C,), 3)).requires_grad_(True)
    gb = (0.1 * rnd((C,), 4)).requires_grad_(True)
    ref = F.gelu(F.group_norm(F.conv1d(x.unsqueeze(1), w, stride=5), C, gw, gb, 1e-5)).transpose(1, 2)   # [B,L0,C]
    dout = rnd(tuple(ref.shape), 5)
    ref.backward(dout)
    d = "cuda"
    out, stats = K.conv0_gn_gelu_fwd(x.to(d), w.detach().to(d), gw.detach().to(d), gb.detach().to(d), dtype)
    close(out, ref, TOL[dtype], "conv0 fwd")
    dw = torch.zeros((C, 1, 10), device=d)
    dgw = torch.zeros((C,), device=d)
    dgb = torch.zeros((C,), device=d)
    K.conv0_gn_gelu_bwd(x.to(d), w.detach().to(d), gw.detach().to(d), gb.detach().to(d), stats, dout.to(dtype).to(d), dw, dgw, dgb)
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    close(dw, w.grad, tol, "conv0 dw")
    close(dgw, gw.grad, tol, "conv0 dgn_w")
    close(dgb, gb.grad, tol, "conv0 dgn_b")


def test_conv0_statistics_on_correlated_audio_and_differencing_filters():
    """GroupNorm statistics come from 65 waveform moments per utterance (sum_t v^2 = w^T R w): the hard case is a smooth,
    DC-shifted waveform against differencing filters, whose outputs are tiny next to the energy of their taps
    (cancellation in w^T R w).  Moments are accumulated in f64, so (mean, rstd) must still match an f64 reference."""
    from audio8_amd import kernels as K
    B, L, C = 2, 40000, 512
    t = torch.arange(L, dtype=torch.float64)
    x = torch.stack([0.3 + 0.5 * torch.sin(2 * math.pi * t / 400.0) + 0.01 * rnd((L,), 1).double(),
                     -0.7 + 0.2 * torch.sin(2 * math.pi * t / 90.0) * torch.cos(2 * math.pi * t / 7000.0) + 1e-3 * rnd((L,), 2).double()]).float()
    w = rnd((C, 1, 10), 3, math.sqrt(2 / 10))
    w[0, 0] = torch.tensor([1., -1., 0, 0, 0, 0, 0, 0, 0, 0])                      # first difference
    w[1, 0] = torch.tensor([1., -2., 1., 0, 0, 0, 0, 0, 0, 0])                     # second difference
    w[2, 0] = torch.tensor([1., -4., 6., -4., 1., 0, 0, 0, 0, 0])                  # fourth difference
    w[3, 0] = 0.1                                                                  # moving average (large mean, small variance)
    w[4:260, 0] -= w[4:260, 0].mean(-1, keepdim=True)                              # zero-DC random filters
    v = F.conv1d(x.double().unsqueeze(1), w.double(), stride=5)                    # [B, C, L0]
    mu, var = v.mean(-1), v.var(-1, unbiased=False)
    gw, gb = torch.ones(C), torch.zeros(C)
    out, stats = K.conv0_gn_gelu_fwd(x.cuda(), w.cuda(), gw.cuda(), gb.cuda(), torch.float32)
    st = stats.cpu().double()
    scale = var.sqrt()
    assert float(((st[..., 0] - mu).abs() / (scale + 1e-3)).max()) < 1e-5
    np.testing.assert_allclose(st[..., 1].numpy(), (1.0 / (var + 1e-5).sqrt()).numpy(), rtol=2e-5)
    ref = F.gelu(F.group_norm(v, C, gw.double(), gb.double(), 1e-5)).transpose(1, 2)
    close(out, ref, 1e-4, "conv0 fwd on correlated audio")


def test_conv0_golden_g3(golden_dir):
    """Layer 0 of the golden conv stack (reference's own ConvFeatureExtractionModel) in f32."""
    from audio8_amd import kernels as K
    from tools.gen_golden import conv_fixture_weights
    from oracle import ref_cpu as R
    g = np.load(os.path.join(golden_dir, "g3_conv.npz"))
    ws, gn_w, gn_b = conv_fixture_weights(int(g["b_seed"]))
    x = torch.from_numpy(g["b_x"])
    sd = {f"feature_extractor.conv_layers.{i}.0.weight": torch.from_numpy(w) for i, w in enumerate(ws)}
    sd["feature_extractor.conv_layers.0.2.weight"] = torch.from_numpy(gn_w)
    sd["feature_extractor.conv_layers.0.2.bias"] = torch.from_numpy(gn_b)
    ref0 = R.conv_feature_extractor(x, sd, return_all=True)[0].transpose(1, 2)
    out, _ = K.conv0_gn_gelu_fwd(x.cuda(), torch.from_numpy(ws[0]).cuda(), torch.from_numpy(gn_w).cuda(), torch.from_numpy(gn_b).cuda(), torch.float32)
    close(out, ref0, 2e-5)
    st = g["b_layer_stats"][0]
    np.testing.assert_allclose(out.double().mean().item(), st[0], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(out.double().abs().mean().item(), st[1], rtol=1e-4)


# ------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("cols", [512, 768, 1024])
def test_layernorm_fwd_bwd(dtype, cols):
    from audio8_amd import kernels as K
    rows = 203
    x = rnd((rows, cols), 1).to(dtype)
    res = rnd((rows, cols), 2).to(dtype)
    gm = (1 + 0.1 * rnd((cols,), 3)).requires_grad_(True)
    bt = (0.1 * rnd((cols,), 4)).requires_grad_(True)
    xs = x.float().requires_grad_(True)
    rs = res.float().requires_grad_(True)
    ref = F.layer_norm(xs + rs, (cols,), gm, bt, 1e-6)
    dy, dy2 = rnd((rows, cols), 5).to(dtype), rnd((rows, cols), 6).to(dtype)
    ref.backward(dy.float() + dy2.float())
    d = "cuda"
    y, mean, rstd = K.layernorm_fwd(x.to(d), res.to(d), gm.detach().to(d), bt.detach().to(d), 1e-6)
    close(y, ref, TOL[dtype], "ln fwd")
    dg, db = torch.zeros((cols,), device=d), torch.zeros((cols,), device=d)
    dx, dres = K.layernorm_bwd(dy.to(d), dy2.to(d), x.to(d), res.to(d), gm.detach().to(d), mean, rstd, dg, db)
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    close(dx, xs.grad, tol, "ln dx")
    close(dres, rs.grad, tol, "ln dres")
    close(dg, gm.grad, tol, "ln dgamma")
    close(db, bt.grad, tol, "ln dbeta")


def test_layernorm_fused_dropout_consistency():
    from audio8_amd import kernels as K
    rows, cols, p = 64, 768, 0.3
    d = "cuda"
    x, res = rnd((rows, cols), 1).to(d), rnd((rows, cols), 2).to(d)
    gm, bt = torch.ones(cols, device=d), torch.zeros(cols, device=d)
    xd = K.dropout(x, p, 77, 5)
    keep = (xd != 0).float()
    assert abs(keep.mean().item() - (1 - p)) < 0.02
    p = round(p * 128) / 128                              # the kernels realise the drop probability in 1/128 steps (7-bit hash fields)
    close(xd, x * keep / (1 - p), 1e-6)
    assert (keep[:, 1:] == keep[:, :-1]).float().mean().item() < 0.75 and (keep[1:] == keep[:-1]).float().mean().item() < 0.75
    y1, m1, r1 = K.layernorm_fwd(x, res, gm, bt, 1e-5, p=p, seed=77, rng_stream=5)
    y2, m2, r2 = K.layernorm_fwd(xd, res, gm, bt, 1e-5)
    close(y1, y2, 1e-6)
    dy = rnd((rows, cols), 3).to(d)
    dg, db = torch.zeros(cols, device=d), torch.zeros(cols, device=d)
    dbias = torch.zeros(x.shape[-1], device=x.device)
    dx1, dres1 = K.layernorm_bwd(dy, None, x, res, gm, m1, r1, dg, db, p=p, seed=77, rng_stream=5, dbias=dbias)
    close(dbias, dx1.float().sum(0), 2e-3 if x.dtype != torch.float32 else 1e-5, "ln fused bias column sum")
    dx2, dres2 = K.layernorm_bwd(dy, None, xd, res, gm, m2, r2, dg, db)
    close(dres1, dres2, 1e-5)
    close(dx1, dx2 * keep / (1 - p), 1e-5)


# ------------------------------------------------------------------ element-wise
@pytest.mark.parametrize("dtype", DT)
def test_elementwise(dtype):
    from audio8_amd import kernels as K
    d = "cuda"
    rows, cols = 98, 768
    x = rnd((rows, cols), 1).to(dtype)
    mask = torch.zeros(rows, dtype=torch.uint8)
    mask[torch.randperm(rows, generator=torch.Generator().manual_seed(0))[:40]] = 1
    emb = rnd((cols,), 2)
    ref = torch.where(mask.bool().unsqueeze(1), emb.to(dtype).unsqueeze(0), x)
    got = K.mask_fill_rows_(x.clone().to(d), mask.to(d), emb.to(d))
    assert torch.equal(got.cpu(), ref)
    dy = rnd((rows, cols), 3).to(dtype)
    demb = torch.zeros(cols, device=d)
    dyo = K.mask_fill_rows_bwd_(dy.clone().to(d), mask.to(d), demb)
    close(demb, dy.float()[mask.bool()].sum(0), 1e-5 if dtype == torch.float32 else 1e-2)
    assert torch.equal(dyo.cpu(), torch.where(mask.bool().unsqueeze(1), torch.zeros_like(dy), dy))
    idx = torch.nonzero(mask).flatten().int()
    gat = K.gather_rows(x.to(d), idx.to(d))
    assert torch.equal(gat.cpu(), x[idx.long()])
    dst = torch.zeros((rows, cols), dtype=dtype, device=d)
    K.scatter_add_rows_(dst, idx.to(d), gat)
    assert torch.equal(dst.cpu(), torch.where(mask.bool().unsqueeze(1), x, torch.zeros_like(x)))
    close(K.add(x.to(d), dy.to(d)), (x.float() + dy.float()).to(dtype), 1e-6)
    close(K.cast(x.to(d), torch.float32), x.float(), 0)
    z = rnd((rows, cols), 4).to(dtype)
    zz = z.float().requires_grad_(True)
    F.gelu(zz).backward(dy.float())
    db = torch.zeros(cols, device=d)
    dz = K.gelu_bwd_bias(dy.to(d), z.to(d), db)
    close(dz, zz.grad, TOL[dtype])
    close(db, zz.grad.sum(0), 1e-4 if dtype == torch.float32 else 3e-2)
    keep = (torch.arange(rows) % 3 != 0).to(torch.uint8)
    zr = K.zero_rows_(x.clone().to(d), keep.to(d))
    assert torch.equal(zr.cpu(), x * keep.bool().unsqueeze(1))
    cm = (torch.arange(2 * cols).reshape(2, cols) % 5 == 0).to(torch.uint8)
    x3 = x[:96].reshape(2, 48, cols)
    zc = K.mask_zero_cols_(x3.clone().to(d), cm.to(d))
    assert torch.equal(zc.cpu(), x3.masked_fill(cm.bool().unsqueeze(1), 0))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("k", [3, 2])
def test_col2im_gelu_grad(dtype, k):
    from audio8_amd import kernels as K
    B, Lin, C, s = 2, 38, 64, 2
    Lout = (Lin - k) // s + 1
    dcol = rnd((B * Lout, k * C), 1).to(dtype)
    z = rnd((B, Lin, C), 2).to(dtype)
    ref = torch.zeros(B, Lin, C)
    dc = dcol.float().view(B, Lout, k, C)
    for t in range(Lout):
        for kk in range(k):
            ref[:, t * s + kk] += dc[:, t, kk]
    zz = z.float().requires_grad_(True)
    F.gelu(zz).sum().backward()
    got = K.col2im_gelu_grad(dcol.cuda(), z.cuda(), B, Lin, Lout, C, k, s)
    close(got, ref * zz.grad, TOL[dtype])


# ------------------------------------------------------------------ positional conv
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("dm,T", [(768, 100), (1024, 300)])
def test_posconv(dtype, dm, T):
    from audio8_amd import kernels as K
    from oracle import ref_cpu as R
    B, G, k = 2, 16, 128
    dg = dm // G
    x = rnd((B, T, dm), 1).to(dtype)
    v = rnd((dm, dg, k), 2, 0.02)
    gnorm = v.norm(2, dim=(0, 1), keepdim=True) * (1 + 0.1 * rnd((1, 1, k), 3))
    bias = 0.1 * rnd((dm,), 4)
    w = R.weight_norm_dim2(gnorm, v).to(dtype).float().requires_grad_(True)     # [dm, dg, k]
    xs = x.float().requires_grad_(True)
    zref = F.conv1d(F.pad(xs.transpose(1, 2), (63, 64)), w, bias, groups=G).transpose(1, 2)
    yref = xs + F.gelu(zref)
    dy = rnd((B, T, dm), 5).to(dtype)
    d = "cuda"
    # kernel weight layout [G][co][kk][ci]
    wk = w.detach().view(G, dg, dg, k).permute(0, 1, 3, 2).contiguous().to(dtype).to(d)
    y, z = K.posconv_fwd(x.to(d), wk, bias.to(d), G, k, 63, 1, want_z=True)
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    close(z, zref, tol, "posconv z")
    close(y, yref, tol, "posconv y")
    # backward through the conv only: dz -> dx (dgrad mode) and dw
    dz = rnd((B, T, dm), 6).to(dtype)
    zref.backward(dz.float())
    wt = w.detach().view(G, dg, dg, k).flip(3).permute(0, 2, 3, 1).contiguous().to(dtype).to(d)   # [G][ci][kk'][co]
    dx, _ = K.posconv_fwd(dz.to(d), wt, None, G, k, k - 1 - 63, -1)
    close(dx, xs.grad, tol, "posconv dgrad")
    dw = torch.zeros((G, dg, k, dg), device=d)
    K.posconv_wgrad(x.to(d), dz.to(d), dw, G, k, 63)
    dw_ref = w.grad.view(G, dg, dg, k).permute(0, 1, 3, 2)
    close(dw, dw_ref, 2e-4 if dtype == torch.float32 else 3e-2, "posconv wgrad")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("dm,k", [(768, 128), (1024, 128), (96, 32)])
def test_weightnorm_pack_and_bwd(dtype, dm, k):
    """a8_weightnorm_pack / a8_weightnorm_bwd against torch's tape through nn.utils.weight_norm(dim=2)'s formula
    (audio8/wav2vec2.py:609): operand images in both layouts, inverse norms, dv and dg (accumulating)."""
    from audio8_amd import kernels as K
    from oracle import ref_cpu as R
    G = 16 if dm >= 768 else 2
    dg = dm // G
    v = rnd((dm, dg, k), 2, 0.02).requires_grad_(True)
    g = (v.detach().norm(2, dim=(0, 1), keepdim=True) * (1 + 0.1 * rnd((1, 1, k), 3))).requires_grad_(True)
    w = R.weight_norm_dim2(g, v)                                                  # [dm, dg, k]
    d = "cuda"
    wf, wt, inv_norm = K.weightnorm_pack(v.detach().to(d), g.detach().to(d), G, dtype)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    close(wf, w.detach().view(G, dg, dg, k).permute(0, 1, 3, 2), tol, "wf")
    close(wt, w.detach().view(G, dg, dg, k).flip(3).permute(0, 2, 3, 1), tol, "wt")
    close(inv_norm, 1.0 / v.detach().norm(2, dim=(0, 1)), 1e-5, "inv_norm")
    if dtype != torch.float32:
        return
    dw = rnd((dm, dg, k), 7)
    w.backward(dw)
    dwg = dw.view(G, dg, dg, k).permute(0, 1, 3, 2).contiguous().to(d)            # [G][co][kk][ci]
    dv0, dg0 = rnd((dm, dg, k), 8), rnd((1, 1, k), 9)
    dv, dgain = dv0.clone().to(d), dg0.clone().to(d)
    K.weightnorm_bwd_(dwg, v.detach().to(d), g.detach().to(d), inv_norm, dv, dgain, G)
    close(dv, dv0 + v.grad, 2e-5, "dv")
    close(dgain, dg0 + g.grad, 2e-5, "dg")


# ------------------------------------------------------------------ attention
def _attn_ref(q, k, v, keymask, scale, drop_keep=None, p=0.0):
    s = torch.matmul(q, k.transpose(-2, -1)) * scale
    if keymask is not None:
        s = s.masked_fill(keymask[:, None, None, :] == 0, -1e9)
    a = s.softmax(-1)
    if drop_keep is not None:
        a = a * drop_keep / (1 - round(p * 256) / 256)       # the kernels realise the drop probability in 1/256 steps
    return torch.matmul(a, v)


@pytest.fixture(params=[(41, 41, 41), (22, 22, 22), (42, 42, 42), (4, 41, 41), (5, 41, 41), (0, 41, 41), (0, 32, 32), (0, 32, 41), (0, 41, 32)], ids=lambda g: "fwd%d-kv%d-q%d" % g)
def attn_geometry(request):
    """Launch geometry (a8_set_tuning keys 2, 3, 4) of the forward, dK/dV and dQ attention kernels: 41 / 22 / 42 = the 16x16-MFMA kernels
    (waves per block * 10 + 16-row groups per wave); forward values below 10 = the 32x32-MFMA kernel for 16-bit data (4 / 5 waves of 32
    queries, 0 = default).  f32 data always runs the 16x16 kernels."""
    from audio8_amd import _lib as L
    for key, v in zip((2, 3, 4), request.param):
        L.check(L.lib().a8_set_tuning(key, v), "a8_set_tuning")
    yield request.param
    for key, v in ((2, ATTN_DEFAULT[0]), (3, ATTN_DEFAULT[1]), (4, ATTN_DEFAULT[2])):
        L.check(L.lib().a8_set_tuning(key, v), "a8_set_tuning")


ATTN_DEFAULT = (0, 41, 41)       # the library's defaults (runtime.hip): 32x32 forward kernel for 16-bit data, 16x16 backward kernels


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("T,use_mask", [(64, False), (100, True), (333, False)])
def test_attention_fwd_bwd(dtype, T, use_mask, attn_geometry):
    from audio8_amd import kernels as K
    B, H = 2, 3
    d = H * 64
    qkv = rnd((B * T, 3 * d), 1).to(dtype)
    dout = rnd((B * T, d), 2).to(dtype)
    keymask = None
    if use_mask:
        keymask = torch.ones(B, T, dtype=torch.uint8)
        keymask[0, 70:] = 0
        keymask[1, 90:] = 0
    f = qkv.float().view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)        # [3,B,H,T,64]
    q, k, v = (f[i].clone().requires_grad_(True) for i in range(3))
    scale = 1 / 8.0
    oref = _attn_ref(q, k, v, keymask, scale)                           # [B,H,T,64]
    oref.backward(dout.float().view(B, T, H, 64).permute(0, 2, 1, 3))
    dev = "cuda"
    g = qkv.to(dev)
    km = keymask.to(dev) if use_mask else None
    o, lse = K.attn_fwd(g[:, :d], g[:, d:2 * d], g[:, 2 * d:], B, T, H, 3 * d, scale, km)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    close(o.view(B, T, H, 64), oref.permute(0, 2, 1, 3), tol, "attn o")
    dbias0 = rnd((3 * d,), 9)                                           # q | k | v bias gradients accumulate (+=) here
    dbias = dbias0.clone().to(dev)
    dq, dk, dv, _ = K.attn_bwd(g[:, :d], g[:, d:2 * d], g[:, 2 * d:], o, dout.to(dev), lse, B, T, H, 3 * d, scale, km, dbias=dbias)
    tolb = 1e-4 if dtype == torch.float32 else 3e-2
    close(dq.reshape(B, T, H, 64), q.grad.permute(0, 2, 1, 3), tolb, "attn dq")
    close(dk.reshape(B, T, H, 64), k.grad.permute(0, 2, 1, 3), tolb, "attn dk")
    close(dv.reshape(B, T, H, 64), v.grad.permute(0, 2, 1, 3), tolb, "attn dv")
    ref_b = torch.cat([x.grad.permute(0, 2, 1, 3).reshape(B * T, d).sum(0) for x in (q, k, v)])     # column sums of dq | dk | dv
    close(dbias, dbias0 + ref_b, tolb, "q/k/v bias gradients from the backward kernels")
    # without the fused bias gradients (what the step runs; the 32 x 32-MFMA backward kernels of round 6 take only this form)
    dq, dk, dv, _ = K.attn_bwd(g[:, :d], g[:, d:2 * d], g[:, 2 * d:], o, dout.to(dev), lse, B, T, H, 3 * d, scale, km)
    close(dq.reshape(B, T, H, 64), q.grad.permute(0, 2, 1, 3), tolb, "attn dq (no dbias)")
    close(dk.reshape(B, T, H, 64), k.grad.permute(0, 2, 1, 3), tolb, "attn dk (no dbias)")
    close(dv.reshape(B, T, H, 64), v.grad.permute(0, 2, 1, 3), tolb, "attn dv (no dbias)")


@pytest.mark.parametrize("dtype", DT)
def test_attention_key_mask_shapes(dtype, attn_geometry):
    """Round 6: a key-padding mask that is a PREFIX of attended keys (what sequence_mask produces, audio8/train.py:33) runs as the unmasked
    kernel over the attended keys -- key tiles beyond the prefix are never staged, dK/dV blocks beyond it store zeros -- and any other mask
    (holes, nothing attended) keeps the per-key path.  Every shape must still equal masked_fill(-1e9) + softmax (audio8's eight_mile
    attention): prefix lengths on, next to and far from the 64-key tile / 64-key block boundaries, one attended key, a full
    row, a row with holes, and a row with no attended key at all (uniform attention over every key, as the reference's -1e9 fill gives)."""
    from audio8_amd import kernels as K
    T, H = 333, 2
    d = H * 64
    rows = []
    for n in (1, 63, 64, 65, 128, 130, 300, 333):
        m = torch.zeros(T, dtype=torch.uint8)
        m[:n] = 1
        rows.append(m)
    holes = torch.ones(T, dtype=torch.uint8)
    holes[10:20] = 0
    holes[200:260] = 0
    rows.append(holes)
    late = torch.zeros(T, dtype=torch.uint8)          # attended keys that do not start at 0: not a prefix
    late[100:180] = 1
    rows.append(late)
    rows.append(torch.zeros(T, dtype=torch.uint8))    # nothing attended
    keymask = torch.stack(rows)
    B = keymask.shape[0]
    qkv = rnd((B * T, 3 * d), 11).to(dtype)
    dout = rnd((B * T, d), 12).to(dtype)
    f = qkv.float().view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    q, k, v = (f[i].clone().requires_grad_(True) for i in range(3))
    oref = _attn_ref(q, k, v, keymask, 0.125)
    oref.backward(dout.float().view(B, T, H, 64).permute(0, 2, 1, 3))
    g = qkv.cuda()
    km = keymask.cuda()
    o, lse = K.attn_fwd(g[:, :d], g[:, d:2 * d], g[:, 2 * d:], B, T, H, 3 * d, 0.125, km)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    tolb = 1e-4 if dtype == torch.float32 else 3e-2
    dq, dk, dv, _ = K.attn_bwd(g[:, :d], g[:, d:2 * d], g[:, 2 * d:], o, dout.cuda(), lse, B, T, H, 3 * d, 0.125, km)
    for b in range(B):                                 # per utterance: an error in one mask shape must not hide in the batch norm
        tag = "mask row %d (%d attended)" % (b, int(keymask[b].sum()))
        close(o.view(B, T, H, 64)[b], oref.permute(0, 2, 1, 3)[b], tol, "o, " + tag)
        close(dq.reshape(B, T, H, 64)[b], q.grad.permute(0, 2, 1, 3)[b], tolb, "dq, " + tag)
        close(dk.reshape(B, T, H, 64)[b], k.grad.permute(0, 2, 1, 3)[b], tolb, "dk, " + tag)
        close(dv.reshape(B, T, H, 64)[b], v.grad.permute(0, 2, 1, 3)[b], tolb, "dv, " + tag)
    # masked keys of a prefix row get EXACTLY zero gradient (their dK/dV blocks are stored, not computed)
    assert float(dk.reshape(B, T, H, 64)[2, 64:].abs().max()) == 0.0 and float(dv.reshape(B, T, H, 64)[2, 64:].abs().max()) == 0.0


def test_attention_bf16_tolerance_detects_a_wrong_scale():
    """What the 16-bit bounds of test_attention_fwd_bwd are worth: the same bf16 kernels run with the softmax scale 10 % off (the kind of
    error a wrong 1 / sqrt(d_k) in one layer would be) must FAIL those bounds in the output and in every gradient, by a clear margin
    over the error the right scale leaves.  The end-to-end bf16
    band (test_fullsize_gpu.py) is wide by design -- cancellation over 12 layers; THIS is where a wrong constant in a bf16-only kernel
    (attn_fwd32, the flagged backward instantiations) is caught."""
    from audio8_amd import kernels as K
    B, H, T = 2, 3, 200
    d = H * 64
    dtype = torch.bfloat16
    qkv = (rnd((B * T, 3 * d), 1) * 2.0).to(dtype)                     # |scores| ~ 2: a softmax that is neither flat nor one-hot
    dout = rnd((B * T, d), 2).to(dtype)
    f = qkv.float().view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    q, k, v = (f[i].clone().requires_grad_(True) for i in range(3))
    scale = 1 / 8.0
    oref = _attn_ref(q, k, v, None, scale)
    oref.backward(dout.float().view(B, T, H, 64).permute(0, 2, 1, 3))
    g = qkv.cuda()

    def run(s_):
        o, lse = K.attn_fwd(g[:, :d], g[:, d:2 * d], g[:, 2 * d:], B, T, H, 3 * d, s_, None)
        dq, dk, dv, _ = K.attn_bwd(g[:, :d], g[:, d:2 * d], g[:, 2 * d:], o, dout.cuda(), lse, B, T, H, 3 * d, s_, None)
        return [(o.view(B, T, H, 64), oref.permute(0, 2, 1, 3), 2e-2), (dq.reshape(B, T, H, 64), q.grad.permute(0, 2, 1, 3), 3e-2),
                (dk.reshape(B, T, H, 64), k.grad.permute(0, 2, 1, 3), 3e-2), (dv.reshape(B, T, H, 64), v.grad.permute(0, 2, 1, 3), 3e-2)]
    def worst(got, ref):                                                 # close()'s own measure: max |got - ref| / (tol-free) scale
        got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
        return float(((got - ref).abs() / (ref.abs() + max(1.0, float(ref.abs().max())))).max())
    right = run(scale)
    for got, ref, tol in right:                                          # the right scale passes ...
        close(got, ref, tol)
    for i, ((got, ref, tol), (g0, r0, _)) in enumerate(zip(run(scale * 1.10), right)):      # ... 10 % off fails every one of the four checks
        print("check %d: error with the right scale %.2e, 10 %% off %.2e (bound %.0e)" % (i, worst(g0, r0), worst(got, ref), tol))
        with pytest.raises(AssertionError):
            close(got, ref, tol, "check %d must notice a 10 %% scale error" % i)
        assert worst(got, ref) > 3 * worst(g0, r0), i


def test_attention_dropout_mask_consistency():
    """Extract the kernel's keep-mask (q = k = 0 -> uniform P; V = I), then check fwd/bwd against a
    reference that applies exactly that mask."""
    from audio8_amd import kernels as K
    B, H, T, p, seed = 1, 2, 64, 0.25, 1234
    d = H * 64
    dev = "cuda"
    z = torch.zeros((B * T, d), device=dev)
    eye = torch.eye(64, device=dev).repeat(1, H)                         # V[key][h*64 + c] = (key == c)
    o, _ = K.attn_fwd(z, z, eye, B, T, H, d, 0.125, None, p=p, seed=seed)
    keep = (o.view(T, H, 64).permute(1, 0, 2) > 0).float().cpu()        # [H, q, key]
    assert abs(keep.mean().item() - (1 - p)) < 0.03
    np.testing.assert_allclose(o.view(T, H, 64)[o.view(T, H, 64) > 0].cpu().numpy(), 1 / 64 / (1 - p), rtol=1e-5)
    qkv = rnd((B * T, 3 * d), 3)
    dout = rnd((B * T, d), 4)
    f = qkv.view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    q, k, v = (f[i].clone().requires_grad_(True) for i in range(3))
    oref = _attn_ref(q, k, v, None, 0.125, keep.unsqueeze(0), p)
    oref.backward(dout.view(B, T, H, 64).permute(0, 2, 1, 3))
    g = qkv.to(dev)
    o, lse = K.attn_fwd(g[:, :d], g[:, d:2 * d], g[:, 2 * d:], B, T, H, 3 * d, 0.125, None, p=p, seed=seed)
    close(o.view(B, T, H, 64), oref.permute(0, 2, 1, 3), 2e-5)
    dq, dk, dv, _ = K.attn_bwd(g[:, :d], g[:, d:2 * d], g[:, 2 * d:], o, dout.to(dev), lse, B, T, H, 3 * d, 0.125, None, p=p, seed=seed)
    close(dq.reshape(B, T, H, 64), q.grad.permute(0, 2, 1, 3), 1e-4)
    close(dk.reshape(B, T, H, 64), k.grad.permute(0, 2, 1, 3), 1e-4)
    close(dv.reshape(B, T, H, 64), v.grad.permute(0, 2, 1, 3), 1e-4)


def _extract_keep(K, B, H, T, p, seed, dtype):
    """The kernel's keep-mask [B,H,q,key]: q = k = 0 -> uniform P; V one-hot over 64-key slabs."""
    d = H * 64
    dev = "cuda"
    z = torch.zeros((B * T, d), device=dev, dtype=dtype)
    keep = torch.zeros(B, H, T, T)
    for k0 in range(0, T, 64):
        v = torch.zeros(B, T, H, 64, device=dev)
        n = min(64, T - k0)
        v[:, k0:k0 + n] = torch.eye(64, device=dev)[:n, None, :]
        o, _ = K.attn_fwd(z, z, v.reshape(B * T, d).to(dtype), B, T, H, d, 0.125, None, p=p, seed=seed)
        keep[:, :, :, k0:k0 + n] = (o.view(B, T, H, 64).permute(0, 2, 1, 3)[..., :n] > 0).float().cpu()
    return keep


@pytest.mark.parametrize("dtype", DT)
def test_attention_dropout_multi_tile(dtype, attn_geometry):
    """Several key/query tiles with a tail (T = 200): forward and both backward kernels must regenerate the same
    keep-mask from the running hash sums."""
    from audio8_amd import kernels as K
    B, H, T, p, seed = 2, 2, 200, 0.1, 77
    d = H * 64
    dev = "cuda"
    keep = _extract_keep(K, B, H, T, p, seed, dtype)
    assert abs(keep.mean().item() - (1 - p)) < 0.01
    # neighbouring rows / keys must not share a mask
    assert (keep[:, :, 1:] == keep[:, :, :-1]).float().mean().item() < 0.9
    assert (keep[..., 1:] == keep[..., :-1]).float().mean().item() < 0.9
    qkv = rnd((B * T, 3 * d), 5).to(dtype)
    dout = rnd((B * T, d), 6).to(dtype)
    f = qkv.float().view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    q, k, v = (f[i].clone().requires_grad_(True) for i in range(3))
    oref = _attn_ref(q, k, v, None, 0.125, keep, p)
    oref.backward(dout.float().view(B, T, H, 64).permute(0, 2, 1, 3))
    g = qkv.to(dev)
    o, lse = K.attn_fwd(g[:, :d], g[:, d:2 * d], g[:, 2 * d:], B, T, H, 3 * d, 0.125, None, p=p, seed=seed)
    tol, tolb = (2e-5, 1e-4) if dtype == torch.float32 else (2e-2, 3e-2)
    close(o.view(B, T, H, 64), oref.permute(0, 2, 1, 3), tol, "attn o (dropout)")
    dq, dk, dv, _ = K.attn_bwd(g[:, :d], g[:, d:2 * d], g[:, 2 * d:], o, dout.to(dev), lse, B, T, H, 3 * d, 0.125, None, p=p, seed=seed)
    close(dq.reshape(B, T, H, 64), q.grad.permute(0, 2, 1, 3), tolb, "attn dq (dropout)")
    close(dk.reshape(B, T, H, 64), k.grad.permute(0, 2, 1, 3), tolb, "attn dk (dropout)")
    close(dv.reshape(B, T, H, 64), v.grad.permute(0, 2, 1, 3), tolb, "attn dv (dropout)")


def test_attention_running_max_rescale(attn_geometry):
    """Scores that grow from key tile to key tile by far more than the lazy-rescale margin (and, for other rows,
    by less than it): the deferred max update must give the exact softmax."""
    from audio8_amd import kernels as K
    B, H, T = 1, 2, 320
    d = H * 64
    torch.manual_seed(11)
    q = torch.randn(B, H, T, 64)
    k = torch.randn(B, H, T, 64) * 0.3
    ramp = torch.linspace(0.0, 6.0, T)                       # key t has a component along u growing with t
    u = torch.nn.functional.normalize(torch.randn(64), dim=0)
    k = k + ramp[None, None, :, None] * u
    q[:, :, ::2] += 40.0 * u                                  # even queries: scores climb ~ 0.125*40*6 = 30 nats across tiles
    q[:, :, 1::2] += 4.0 * u                                  # odd queries: ~3 nats, inside the margin
    v = torch.randn(B, H, T, 64)
    qkv = torch.stack([q, k, v], 0).permute(1, 3, 0, 2, 4).reshape(B * T, 3 * d)   # [B,T,3,H,64]
    dout = rnd((B * T, d), 8)
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    oref = _attn_ref(qr, kr, vr, None, 0.125)
    oref.backward(dout.view(B, T, H, 64).permute(0, 2, 1, 3))
    g = qkv.to("cuda")
    o, lse = K.attn_fwd(g[:, :d], g[:, d:2 * d], g[:, 2 * d:], B, T, H, 3 * d, 0.125, None)
    close(o.view(B, T, H, 64), oref.permute(0, 2, 1, 3), 3e-5, "attn o (rescale)")
    lse_ref = torch.logsumexp(torch.matmul(q, k.transpose(-2, -1)) * 0.125, -1) * 1.4426950408889634
    close(lse.view(B, H, T), lse_ref, 1e-5, "attn lse (rescale)")
    dq, dk, dv, _ = K.attn_bwd(g[:, :d], g[:, d:2 * d], g[:, 2 * d:], o, dout.to("cuda"), lse, B, T, H, 3 * d, 0.125, None)
    close(dq.reshape(B, T, H, 64), qr.grad.permute(0, 2, 1, 3), 1e-4, "attn dq (rescale)")
    close(dk.reshape(B, T, H, 64), kr.grad.permute(0, 2, 1, 3), 1e-4, "attn dk (rescale)")
    close(dv.reshape(B, T, H, 64), vr.grad.permute(0, 2, 1, 3), 1e-4, "attn dv (rescale)")


@pytest.mark.parametrize("nw", [4, 5])
def test_attention_fwd32_rescale_paths_bf16(nw):
    """The 16-bit forward kernel has no running-max pass: probabilities are referred to the max of the last rescale and the tile's row
    sum is the overflow test (> 2^8 per lane -> rescale path).  Inputs that FORCE each branch at chosen tiles, checked over the full
    tensor against an fp64 softmax of the same bf16 inputs: scores that jump far past the limit (even queries: +30 nats over the
    sequence, and one spiked key in the third tile), scores that creep up inside the margin (odd queries), a key mask, a ragged tail
    (T = 333: last tile has 13 keys), and dropout on top (mask extracted from the kernel itself)."""
    from audio8_amd import kernels as K, _lib as L
    B, H, T = 2, 2, 333
    d = H * 64
    torch.manual_seed(5)
    q = torch.randn(B, H, T, 64)
    k = torch.randn(B, H, T, 64) * 0.3
    ramp = torch.linspace(0.0, 6.0, T)
    u = torch.nn.functional.normalize(torch.randn(64), dim=0)
    k = k + ramp[None, None, :, None] * u
    k[:, :, 150] += 9.0 * u                                  # a spike inside tile 2: + 0.125 * 40 * 9 = 45 nats for the even queries
    q[:, :, ::2] += 40.0 * u
    q[:, :, 1::2] += 4.0 * u
    v = torch.randn(B, H, T, 64)
    bf = torch.bfloat16
    qkv = torch.stack([q, k, v], 0).permute(1, 3, 0, 2, 4).reshape(B * T, 3 * d).to(bf)
    f = qkv.double().view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    qd, kd, vd = f[0], f[1], f[2]
    keymask = torch.ones(B, T, dtype=torch.uint8)
    keymask[1, 300:] = 0
    g = qkv.to("cuda")
    L.check(L.lib().a8_set_tuning(2, nw), "tune")
    try:
        for km in (None, keymask):
            sc = torch.matmul(qd, kd.transpose(-2, -1)) * 0.125
            if km is not None:
                sc = sc.masked_fill(km[:, None, None, :] == 0, -1e9)
            oref = torch.matmul(sc.softmax(-1), vd)
            o, lse = K.attn_fwd(g[:, :d], g[:, d:2 * d], g[:, 2 * d:], B, T, H, 3 * d, 0.125, None if km is None else km.cuda())
            close(o.float().view(B, T, H, 64), oref.float().permute(0, 2, 1, 3), 2e-2, "fwd32 o")
            close(lse.view(B, H, T), (torch.logsumexp(sc, -1) * 1.4426950408889634).float(), 2e-3, "fwd32 lse")
        # dropout: the kernel's own mask, extracted with uniform probabilities (same hash words as the 16x16 kernels)
        p, seed = 0.1, 91
        keep = _extract_keep(K, B, H, T, p, seed, bf)
        L.check(L.lib().a8_set_tuning(2, 41), "tune")
        keep_old = _extract_keep(K, B, H, T, p, seed, bf)
        L.check(L.lib().a8_set_tuning(2, nw), "tune")
        assert torch.equal(keep, keep_old)                    # backward (16x16 kernels) regenerates exactly this mask
        sc = torch.matmul(qd, kd.transpose(-2, -1)) * 0.125
        oref = torch.matmul(sc.softmax(-1) * keep.double() / (1 - round(p * 256) / 256), vd)
        o, _ = K.attn_fwd(g[:, :d], g[:, d:2 * d], g[:, 2 * d:], B, T, H, 3 * d, 0.125, None, p=p, seed=seed)
        close(o.float().view(B, T, H, 64), oref.float().permute(0, 2, 1, 3), 2e-2, "fwd32 o (dropout)")
    finally:
        L.check(L.lib().a8_set_tuning(2, ATTN_DEFAULT[0]), "tune")


# ------------------------------------------------------------------ VQ (golden G4)
def test_vq_against_golden(golden_dir):
    from audio8_amd import kernels as K, _lib as L
    g = np.load(os.path.join(golden_dir, "g4_vq.npz"))
    d = "cuda"
    B, Tm, Cin = g["y"].shape
    G, V, D = 2, 320, 128
    N = B * Tm
    y = torch.from_numpy(g["y"]).view(N, Cin).to(d)
    wp, bp = torch.from_numpy(g["wp"]).to(d), torch.from_numpy(g["bp"]).to(d)
    vars2d = torch.from_numpy(g["vars"]).view(G * V, D).to(d)
    logits = K.gemm(L.NT, y, wp, N, G * V, Cin, bias=bp, out_dtype=torch.float32).view(N * G, V)
    # eval: argmax
    codes, avg = K.vq_select(logits, None, 0.5)
    q = K.vq_gather(vars2d, codes, N, G, V, D, torch.float32)
    ppl, dppl = K.vq_ppl(avg)
    close(q.view(B, Tm, G * D), torch.from_numpy(g["eval_q"]), 1e-6)
    np.testing.assert_allclose(ppl.item(), g["eval_ppl"], rtol=1e-5)
    # train: injected gumbel noise
    gum = torch.from_numpy(g["gumbel"]).to(d)
    codes, avg = K.vq_select(logits, gum, 0.5)
    q = K.vq_gather(vars2d, codes, N, G, V, D, torch.float32)
    ppl, dppl = K.vq_ppl(avg)
    close(q.view(B, Tm, G * D), torch.from_numpy(g["train_q"]), 1e-6)
    np.testing.assert_allclose(ppl.item(), g["train_ppl"], rtol=1e-5)
    # backward of sum(q * gq): no diversity term
    gq = torch.from_numpy(g["train_gq"]).view(N, G * D).to(d)
    eproj = torch.empty((N * G, V), device=d)
    for gi in range(G):
        K.gemm(L.NT, gq[:, gi * D:], vars2d[gi * V:], N, V, D, lda=G * D, out=eproj.view(N, G * V)[:, gi * V:], ldc=G * V)
    dvars = torch.zeros_like(vars2d)
    zero = torch.zeros(1, device=d)
    dlogits = K.vq_bwd(logits, gum, 0.5, codes, vars2d, gq, eproj, dppl, zero, dvars, N, G, V, D, True)
    close(dvars.view(1, G * V, D), torch.from_numpy(g["train_gvars"]), 1e-5)
    dl = dlogits.view(N, G * V)
    gwp = K.gemm(L.TN, dl, y, G * V, Cin, N, out_dtype=torch.float32)
    gy = K.gemm(L.NN, dl, wp, N, Cin, G * V)
    close(gwp, torch.from_numpy(g["train_gwp"]), 1e-4)
    close(dl.sum(0), torch.from_numpy(g["train_gbp"]), 1e-4)
    close(gy.view(B, Tm, Cin), torch.from_numpy(g["train_gy"]), 1e-4)
    # gradient of prob_ppl alone (diversity path)
    one = torch.ones(1, device=d)
    dvars2 = torch.zeros_like(vars2d)
    dl2 = K.vq_bwd(logits, gum, 0.5, codes, vars2d, torch.zeros_like(gq), torch.zeros_like(eproj), dppl, one, dvars2, N, G, V, D, True)
    gy2 = K.gemm(L.NN, dl2.view(N, G * V), wp, N, Cin, G * V)
    close(gy2.view(B, Tm, Cin), torch.from_numpy(g["ppl_gy"]), 1e-4)


# ------------------------------------------------------------------ contrastive (golden G5)
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("group_mode", ["scatter", "grouped", "grouped_wrong_size"])
def test_contrastive_against_golden(golden_dir, dtype, group_mode):
    from audio8_amd import kernels as K
    g = np.load(os.path.join(golden_dir, "g5_loss.npz"))
    d = "cuda"
    mask = torch.from_numpy(g["mask"])
    outputs = torch.from_numpy(g["outputs"])
    B, T, C = outputs.shape
    ctx = outputs[mask.unsqueeze(-1).expand_as(outputs)].view(-1, C)          # [B*Tm, C], b-major
    lat = torch.from_numpy(g["latents"]).view(-1, C)
    Tm = lat.shape[0] // B
    idx = torch.from_numpy(g["idx"]).view(B, Tm, -1).reshape(B * Tm, -1).int()   # [B, N*Tm] is (t, n) row-major per b
    N = idx.shape[1]
    logits, loss_sum, inv_norms = K.contrastive_fwd(ctx.to(dtype).to(d), lat.to(dtype).to(d), idx.to(d))
    ref_logits = torch.from_numpy(g["logits"]).view(Tm, B, N + 1).transpose(0, 1).reshape(B * Tm, N + 1)   # reference rows are t*B + b
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    close(logits, ref_logits, tol, "cosine logits")
    xe = loss_sum.item() / (B * Tm)
    loss = 0.1 * xe + 10 * (640 - float(g["gs"])) / 640
    np.testing.assert_allclose(loss, float(g["loss"]), rtol=1e-5 if dtype == torch.float32 else 5e-3)
    gscale = torch.full((1,), 0.1 / (B * Tm), device=d)
    # "grouped": the latent gradient is accumulated per utterance in LDS; a deliberately wrong group size (rows that
    # point outside their group) must still give the right answer through the per-pair fallback
    rpg = {"scatter": 0, "grouped": Tm, "grouped_wrong_size": max(1, Tm // 2) if (B * Tm) % max(1, Tm // 2) == 0 else 1}[group_mode]
    # the forward's inverse norms (as the model passes them), or recomputed by the backward (scatter mode here)
    dctx, dlat = K.contrastive_bwd(ctx.to(dtype).to(d), lat.to(dtype).to(d), idx.to(d), logits, gscale, rows_per_group=rpg,
                                   inv_norms=None if group_mode == "scatter" else inv_norms)
    g_out = torch.from_numpy(g["g_outputs"])
    ref_dctx = g_out[mask.unsqueeze(-1).expand_as(g_out)].view(-1, C)
    tolb = 1e-4 if dtype == torch.float32 else 3e-2
    close(dctx, ref_dctx, tolb, "dctx")
    close(dlat, torch.from_numpy(g["g_latents"]).view(-1, C), tolb, "dlat")


# ------------------------------------------------------------------ CTC (golden G6 + f64 oracle)
@pytest.mark.parametrize("tag", ["a", "b"])
def test_ctc_against_golden_and_oracle(golden_dir, tag):
    from audio8_amd import kernels as K
    from oracle import ref_cpu as R
    g = np.load(os.path.join(golden_dir, "g6_ctc.npz"))
    lp = torch.from_numpy(g[f"{tag}_lp"])
    il, tl, targets = g[f"{tag}_il"], g[f"{tag}_tl"], g[f"{tag}_targets"]
    flat = R.ctc_strip_targets(targets).astype(np.int32)
    off = np.concatenate([[0], np.cumsum(tl)]).astype(np.int32)
    d = "cuda"
    nll, grad = K.ctc_fwd_bwd(lp.to(d), torch.from_numpy(flat).to(d), torch.from_numpy(off).to(d), torch.from_numpy(il.astype(np.int32)).to(d),
                              int(tl.max()))
    want_nll, want_grad = R.ctc_loss_np(g[f"{tag}_lp"], il, targets, tl, reduction="none")
    np.testing.assert_allclose(nll.cpu().numpy(), want_nll, rtol=2e-5)
    np.testing.assert_allclose(grad.cpu().numpy(), want_grad, rtol=1e-3, atol=2e-5)     # vs the f64 oracle
    np.testing.assert_allclose(nll.sum().item(), g[f"{tag}_sum_loss"], rtol=2e-5)        # vs the reference (ATen f32)
    atol = 1e-5 if tag == "a" else 2.5e-3   # the golden's own f32 log-space noise at T=624, see test_oracle_golden
    np.testing.assert_allclose(grad.cpu().numpy(), g[f"{tag}_sum_grad"], rtol=1e-3, atol=atol)


@pytest.mark.parametrize("case", ["small-v", "chunk-edges", "one-frame-chunks", "big-v", "states-per-lane"])
def test_ctc_random_shapes_against_the_f64_oracle(case):
    """The round-6 CTC kernels (alpha / beta on one or four waves per utterance and direction with LDS-staged rows, gradient as a parallel launch)
    over the shapes their staging distinguishes, against the float64 restatement of F.ctc_loss (audio8/ctc.py:186-206): vocabularies that put
    1 .. 64 frames in a 1024-float chunk, utterance lengths on and next to chunk boundaries, V > 1024 (one frame at a time, one- and four-wave),
    every <states per thread, waves> instantiation (2 S + 1 = 1 .. 1023) incl. label counts whose last two states sit in DIFFERENT waves
    (2 S + 1 = 129, 257, 385, 577, 769), repeated labels, empty targets, ragged and zero input lengths, infeasible rows."""
    from audio8_amd import kernels as K
    from oracle import ref_cpu as R
    rs = np.random.RandomState({"small-v": 1, "chunk-edges": 2, "one-frame-chunks": 3, "big-v": 4, "states-per-lane": 5}[case])
    if case == "small-v":
        configs = [(rs.randint(1, 120), rs.randint(1, 7), V, rs.randint(0, 20)) for V in (3, 5, 16, 17, 33, 100)]
    elif case == "chunk-edges":                       # V = 32: 32 frames per chunk; V = 16: 64; lengths around multiples of the chunk
        configs = [(T, 3, 32, 9) for T in (31, 32, 33, 64, 65, 96)] + [(T, 2, 16, 5) for T in (63, 64, 65, 129)]
    elif case == "one-frame-chunks":                  # 512 < V <= 1024: one frame per chunk
        configs = [(rs.randint(3, 40), 2, V, 6) for V in (513, 700, 1024)]
    elif case == "big-v":                             # V > 1024: the synchronous path
        configs = [(rs.randint(3, 30), 2, V, 5) for V in (1025, 1500, 3000)] + [(230, 2, 1500, 100)]
    else:
        configs = [(2 * S + 6, 2, 12, S) for S in (0, 1, 31, 32, 33, 63, 64, 100, 128, 160, 192, 255, 288, 300, 384, 448, 511)]
    for T, B, V, Smax in configs:
        lp = torch.log_softmax(torch.from_numpy(rs.standard_normal((T, B, V)).astype(np.float32)) * 2.0, -1)
        tl = rs.randint(0, Smax + 1, B)
        tl[0] = Smax                                   # (the launch's states-per-lane count follows the longest target)
        il = rs.randint(0, T + 1, B)
        il[0] = T
        if B > 1:
            il[-1] = min(T, max(0, int(tl[-1]) - 1))   # too short for its target: infeasible (zero_infinity)
        targets = np.full((B, Smax + 2), 1, dtype=np.int64)
        for b in range(B):
            lab = rs.randint(1, V, int(tl[b]))
            if tl[b] > 3:
                lab[1] = lab[0]                        # a repeated label (no skip transition across it)
            lab[lab == 1] = min(3, V - 1) if V > 2 else 1
            lab[lab == 2] = min(3, V - 1) if V > 3 else lab[lab == 2]
            targets[b, :tl[b]] = lab
            targets[b, tl[b]] = 2
        if V <= 3:
            continue
        flat = R.ctc_strip_targets(targets).astype(np.int32)
        assert flat.size == int(tl.sum())
        off = np.concatenate([[0], np.cumsum(tl)]).astype(np.int32)
        nll, grad = K.ctc_fwd_bwd(lp.cuda(), torch.from_numpy(flat).cuda(), torch.from_numpy(off).cuda(), torch.from_numpy(il.astype(np.int32)).cuda(),
                                  int(max(1, tl.max())))
        want_nll, want_grad = R.ctc_loss_np(lp.numpy(), il, targets, tl, reduction="none")
        tag = "T=%d B=%d V=%d Smax=%d il=%s tl=%s" % (T, B, V, Smax, il.tolist(), tl.tolist())
        np.testing.assert_allclose(nll.cpu().numpy(), want_nll, rtol=3e-5, atol=1e-5, err_msg=tag)
        np.testing.assert_allclose(grad.cpu().numpy(), want_grad, rtol=1e-3, atol=3e-5, err_msg=tag)


@pytest.mark.parametrize("dtype", DT)
def test_log_softmax(dtype):
    from audio8_amd import kernels as K
    x = rnd((300, 32), 1, 3.0).to(dtype)
    xs = x.float().requires_grad_(True)
    ref = F.log_softmax(xs, -1)
    dy = rnd((300, 32), 2)
    ref.backward(dy)
    y = K.log_softmax_fwd(x.cuda())
    close(y, ref, 1e-5)
    dx = K.log_softmax_bwd(y, dy.cuda(), dtype)
    close(dx, xs.grad, TOL[dtype])


# ------------------------------------------------------------------ optimizer
def test_sumsq_adamw_matches_torch():
    from audio8_amd import kernels as K
    n = 100003
    p0 = rnd((n + 1,), 1)[:n]
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p], lr=2e-4, weight_decay=1e-2)
    d = "cuda"
    flat = torch.zeros(n + 5, device=d)[:n]
    flat.copy_(p0)
    m, v = torch.zeros(n, device=d), torch.zeros(n, device=d)
    for step in range(1, 4):
        gr = rnd((n,), 10 + step, 0.05)
        p.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_([p], 1.0)
        opt.step()
        gsq = torch.zeros(1, device=d)
        gd = gr.to(d)
        K.sumsq_(gd, gsq)
        np.testing.assert_allclose(gsq.item(), (gr.double() ** 2).sum().item(), rtol=1e-5)
        K.adamw_(flat, gd, m, v, 2e-4, 0.9, 0.999, 1e-8, 1e-2, step, gsq, 1.0)
        close(flat, p.data, 2e-6)


# ------------------------------------------------------------------ greedy CTC decode (metrics path)
@pytest.mark.parametrize("dtype", DT)
def test_ctc_greedy_decode_matches_host(dtype):
    """a8_ctc_greedy_decode == argmax -> unique_consecutive -> drop blanks per utterance (audio8/ctc.py:156-159), incl.
    runs that straddle the 64-frame chunks, all-blank rows, ragged lengths, ties, and ctc_metrics end to end."""
    from audio8_amd import kernels as K, ctc as Cc
    torch.manual_seed(5)
    B, T, V = 6, 333, 32
    # piecewise-constant label runs so that repeats and blanks are frequent
    lab = torch.randint(0, V, (B, T // 3 + 1)).repeat_interleave(3, dim=1)[:, :T]
    lab[lab % 3 == 0] = 0                                   # many blanks
    lab[1] = 0                                              # an all-blank utterance
    lp = torch.randn(B, T, V) * 0.1
    lp.scatter_(2, lab.unsqueeze(-1), 3.0)
    lp[2, 5, :] = 1.0                                        # a full tie: argmax must pick label 0 (first maximum)
    lp = torch.log_softmax(lp, -1).to(dtype)
    lens = torch.tensor([T, T, 200, 64, 65, 1])
    tok, n = K.ctc_greedy_decode(lp.cuda(), lens.to(torch.int32).cuda(), 0)
    tok, n = tok.cpu(), n.cpu()
    for b in range(B):
        ref = lp[b, :lens[b]].float().argmax(-1).unique_consecutive()
        ref = ref[ref != 0].tolist()
        assert tok[b, :n[b]].tolist() == ref, b
    # metrics: device path == host path, with the reference's sample-length quirk (lengths far beyond T)
    vocab = {i: (c if i >= 4 else "") for i, c in enumerate(["<s>", "<pad>", "</s>", "<unk>"] + list("|abcdefghijklmnopqrstuvwxyz'"))}
    target = torch.randint(4, V, (B, 40))
    target[:, 30:] = Cc.Offsets.PAD
    sample_lens = torch.full((B,), 100000)
    m_host = Cc.ctc_metrics(lp.float(), target, sample_lens, vocab)
    m_dev = Cc.ctc_metrics(lp.cuda(), target.cuda(), sample_lens.cuda(), vocab)
    assert m_host == m_dev and m_host["c_total"] == B * 30


# ------------------------------------------------------------------ in-kernel Gumbel noise, conv weight-gradient fold, memset
def test_vq_select_draws_its_own_gumbel_noise():
    """F.gumbel_softmax adds -log(E), E ~ Exp(1) (torch/nn/functional.py:2163; audio8/wav2vec2.py:556).  With gumbel_gen the select kernel
    draws that noise itself from a counter hash: check its distribution (mean = Euler's gamma, variance = pi^2 / 6, CDF at three
    points), that the stored noise is what the argmax used, and that the seed changes it."""
    from audio8_amd import kernels as K
    R, V = 2944 * 2, 320
    logits = rnd((R, V), 3).cuda()
    g = torch.empty_like(logits)
    codes, avg = K.vq_select(logits, None, 0.5, gumbel_gen=g, seed=123)
    gc = g.double().cpu()
    assert torch.isfinite(gc).all()
    assert abs(gc.mean().item() - 0.5772) < 5e-3 and abs(gc.var().item() - 1.6449) < 2e-2
    for x in (-1.0, 0.0, 1.5):                                   # CDF of Gumbel(0, 1): exp(-exp(-x))
        assert abs((gc < x).double().mean().item() - float(np.exp(-np.exp(-x)))) < 3e-3
    assert torch.equal(codes.cpu().long(), (logits + g).argmax(-1).cpu())
    codes2, avg2 = K.vq_select(logits, g, 0.5)                   # handing the same noise back in reproduces the codes
    assert torch.equal(codes, codes2) and torch.equal(avg, avg2)
    g2 = torch.empty_like(logits)
    K.vq_select(logits, None, 0.5, gumbel_gen=g2, seed=124)
    assert (g2 != g).float().mean().item() > 0.99
    # every value the 23-bit uniform can take gives finite noise: the extremes are -log(-log(1 - 2^-24)) = 16.6 and -log(-log(2^-24)) = -2.8
    big = torch.empty((40000, 320), device="cuda")
    K.vq_select(torch.zeros_like(big), None, 0.5, gumbel_gen=big, seed=7)          # 12.8 M draws: the 24-bit form hit its +inf about once
    assert torch.isfinite(big).all() and big.max().item() < 16.7 and big.min().item() > -2.9
    # NaN logits must not produce an out-of-range code
    bad = torch.full((16, 320), float("nan"), device="cuda")
    c_bad, _ = K.vq_select(bad, None, 0.5)
    assert int(c_bad.min()) >= 0 and int(c_bad.max()) < 320
    # neighbouring elements are uncorrelated
    assert abs(float(np.corrcoef(gc[:, :-1].reshape(-1)[:200000].numpy(), gc[:, 1:].reshape(-1)[:200000].numpy())[0, 1])) < 0.01


@pytest.mark.parametrize("k", [2, 3])
def test_conv_wgrad_fold(k):
    from audio8_amd import kernels as K
    co, cin = 512, 512
    dw2 = rnd((co, k * cin), 4).cuda()
    grad0 = rnd((co, cin, k), 5).cuda()
    grad = grad0.clone()
    K.conv_wgrad_fold_(dw2.clone(), grad, co, cin, k)
    want = grad0 + dw2.view(co, k, cin).permute(0, 2, 1)
    assert torch.equal(grad, want)
    buf = dw2.clone()
    K.conv_wgrad_fold_(buf, grad, co, cin, k)
    assert float(buf.abs().max()) == 0.0                          # cleared behind the read: the caller keeps it across steps


def test_zeros_helper():
    from audio8_amd import kernels as K
    t = K.zeros((3, 1001), torch.float32, "cuda")
    assert t.shape == (3, 1001) and float(t.abs().max()) == 0.0
    e = K.zeros((0, 5), torch.bfloat16, "cuda")
    assert e.numel() == 0
