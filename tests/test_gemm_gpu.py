"""GPU parity of a8_gemm against a CPU fp64 matmul (the oracle for a plain contraction)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(shape, dtype, seed, integer=False):
    g = torch.Generator().manual_seed(seed)
    if integer:
        t = torch.randint(-3, 4, shape, generator=g).float()
    else:
        t = torch.randn(shape, generator=g)
    return t.to(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("layout", ["NT", "NN", "TN"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 104), (781, 768, 512), (48, 32, 2056), (130, 264, 72)])
def test_gemm_layouts_exact_integers(dtype, layout, M, N, K):
    """Small-integer operands: every product and partial sum is exact in bf16-in/f32-acc and f32,
    so any wrong lane map / k-slot / transposed read shows up as a hard mismatch."""
    from audio8_amd import kernels as Kn, _lib as L
    if layout == "TN" and (M % 8 or N % 8):
        M, N = (M + 7) // 8 * 8, (N + 7) // 8 * 8
    if layout == "NN" and N % 8:
        N = (N + 7) // 8 * 8
    A = _mk((M, K), dtype, 1, integer=True)
    B = _mk((N, K), dtype, 2, integer=True)
    ref = A.double() @ B.double().T
    dev = "cuda"
    if layout == "NT":
        out = Kn.gemm(L.NT, A.to(dev), B.to(dev), M, N, K)
    elif layout == "NN":
        out = Kn.gemm(L.NN, A.to(dev), B.T.contiguous().to(dev), M, N, K)
    else:
        out = Kn.gemm(L.TN, A.T.contiguous().to(dev), B.T.contiguous().to(dev), M, N, K, out_dtype=torch.float32)
    torch.cuda.synchronize()
    got = out.cpu().double()
    if dtype == torch.bfloat16 and layout != "TN":
        ref = ref.to(torch.bfloat16).double()   # output rounding only
    assert torch.equal(got, ref), f"max diff {(got - ref).abs().max().item()}"


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 1.5e-2)])
def test_gemm_epilogues(dtype, tol):
    from audio8_amd import kernels as Kn, _lib as L
    M, N, K = 300, 256, 192
    A, B = _mk((M, K), dtype, 3), _mk((N, K), dtype, 4)
    bias = _mk((N,), torch.float32, 5)
    z_ref = (A.double() @ B.double().T) / math.sqrt(K) + bias.double()
    dev = "cuda"
    zout = torch.empty((M, N), device=dev, dtype=dtype)
    out = Kn.gemm(L.NT, A.to(dev), B.to(dev), M, N, K, bias=bias.to(dev), act=L.ACT_GELU, zout=zout, alpha=1 / math.sqrt(K))
    np.testing.assert_allclose(zout.float().cpu().numpy(), z_ref.float().numpy(), rtol=tol, atol=tol)
    np.testing.assert_allclose(out.float().cpu().numpy(), torch.nn.functional.gelu(z_ref).float().numpy(), rtol=tol, atol=tol)
    # dgrad with fused gelu'(z): dx = (dy . W) * gelu'(z)
    dY = _mk((M, N), dtype, 6)
    W = _mk((N, K), dtype, 7)
    Z = _mk((M, K), dtype, 8)
    zz = Z.double().requires_grad_(True)
    torch.nn.functional.gelu(zz).sum().backward()
    ref = (dY.double() @ W.double()) / math.sqrt(N) * zz.grad
    got = Kn.gemm(L.NN, dY.to(dev), W.to(dev), M, K, N, act=L.ACT_MUL_GELU_GRAD, zin=Z.to(dev), ldz=K, alpha=1 / math.sqrt(N))
    np.testing.assert_allclose(got.float().cpu().numpy(), ref.float().numpy(), rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_splitk_wgrad_and_accumulate(dtype):
    from audio8_amd import kernels as Kn, _lib as L
    Mred, N, K = 1000, 64, 136          # dW[N,K] = dY^T[N,Mred] . X[Mred,K]
    dY, X = _mk((Mred, N), dtype, 9, integer=True), _mk((Mred, K), dtype, 10, integer=True)
    ref = dY.double().T @ X.double()
    dev = "cuda"
    out = torch.ones((N, K), device=dev, dtype=torch.float32)
    Kn.gemm(L.TN, dY.to(dev), X.to(dev), N, K, Mred, out=out, accumulate=True, splitk=5)
    assert torch.equal(out.cpu().double(), ref + 1.0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_as_gemm_row_remap(dtype):
    """Strided Conv1d(C->C, k=3, s=2) over a channels-last activation as one NT GEMM over overlapping rows."""
    from audio8_amd import kernels as Kn, _lib as L
    Bn, Lin, Cc, k, s = 3, 41, 64, 3, 2
    Lout = (Lin - k) // s + 1
    x = _mk((Bn, Lin, Cc), dtype, 11, integer=True)
    w = _mk((Cc, Cc, k), dtype, 12, integer=True)                # [Cout, Cin, k]
    ref = torch.nn.functional.conv1d(x.double().transpose(1, 2), w.double(), stride=s).transpose(1, 2)   # [B,Lout,C]
    w2 = w.permute(0, 2, 1).contiguous().view(Cc, k * Cc)        # [Cout][kk*C + ci]
    dev = "cuda"
    out = Kn.gemm(L.NT, x.to(dev), w2.to(dev), Bn * Lout, Cc, k * Cc, lda=s * Cc, a_rpb=Lout, a_bs=Lin * Cc)
    got = out.view(Bn, Lout, Cc).cpu().double()
    if dtype == torch.bfloat16:
        ref = ref.to(torch.bfloat16).double()
    assert torch.equal(got, ref)


def test_gemm_rejects_bad_args():
    from audio8_amd import kernels as Kn, _lib as L
    A = torch.zeros((16, 12), device="cuda", dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="multiple of"):
        Kn.gemm(L.NT, A, A, 16, 16, 12)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("out_f32", [False, True])
def test_gemm_colsum_epilogue(dtype, out_f32):
    """Column sums of the final output values (bias gradient riding on a data-gradient GEMM), with the gelu' epilogue."""
    from audio8_amd import kernels as Kn, _lib as L
    M, N, K = 333, 264, 96
    A, B = _mk((M, K), dtype, 11), _mk((N, K), dtype, 12)
    z = _mk((M, N), dtype, 13)
    dev = "cuda"
    cs = torch.full((N,), 0.5, device=dev)
    out = Kn.gemm(L.NT, A.to(dev), B.to(dev), M, N, K, act=L.ACT_MUL_GELU_GRAD, zin=z.to(dev), colsum=cs,
                  out_dtype=torch.float32 if out_f32 else dtype)
    zz = z.double()
    gp = 0.5 * (1 + torch.erf(zz / math.sqrt(2))) + zz * torch.exp(-0.5 * zz * zz) / math.sqrt(2 * math.pi)
    ref = (A.double() @ B.double().T) * gp
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    np.testing.assert_allclose(out.double().cpu().numpy(), ref.numpy(), rtol=tol, atol=tol * 10)
    np.testing.assert_allclose(cs.double().cpu().numpy(), 0.5 + ref.sum(0).numpy(), rtol=tol, atol=tol * 30)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("layout", ["TN", "NT"])
def test_gemm_grouped(dtype, layout):
    """Several problems of different shapes in one launch == the same problems launched one by one."""
    from audio8_amd import kernels as Kn, _lib as L
    dev = "cuda"
    shapes = [(768, 264, 1000), (128, 128, 72), (200, 512, 2056), (8, 8, 8)]
    probs, refs = [], []
    for i, (M, N, K) in enumerate(shapes):
        A, B = _mk((M, K), dtype, 20 + i, integer=True), _mk((N, K), dtype, 40 + i, integer=True)
        refs.append(A.double() @ B.double().T + 1.0)
        out = torch.ones((M, N), device=dev, dtype=torch.float32)
        if layout == "TN":
            probs.append((A.T.contiguous().to(dev), B.T.contiguous().to(dev), M, N, K, dict(out=out, accumulate=True, splitk=1 + i % 2)))
        else:
            probs.append((A.to(dev), B.to(dev), M, N, K, dict(out=out, accumulate=True)))
    outs = Kn.gemm_grouped(L.TN if layout == "TN" else L.NT, probs)
    for o, r in zip(outs, refs):
        np.testing.assert_array_equal(o.double().cpu().numpy(), r.numpy())


def test_gemm_grouped_asymmetric_ring_matches_default():
    """The grouped launch picks the asymmetric-ring main loop (a8_set_tuning key 14) when every problem is at least that tall:
    bit-identical to the default grouped kernel, M tails and the gelu' epilogue included."""
    from audio8_amd import kernels as Kn, _lib as L
    dev, bf = "cuda", torch.bfloat16
    shapes = [(1000, 512, 1024), (777, 512, 1536), (130, 256, 64)]
    ops_ = [(_mk((M, K), bf, 50 + i).to(dev), _mk((N, K), bf, 60 + i).to(dev), _mk((M, N), bf, 70 + i).to(dev)) for i, (M, N, K) in enumerate(shapes)]

    def run():
        probs = [(A, B, M, N, K, dict(act=L.ACT_MUL_GELU_GRAD, zin=z) if i == 1 else {}) for i, ((M, N, K), (A, B, z)) in enumerate(zip(shapes, ops_))]
        return [o.clone() for o in Kn.gemm_grouped(L.NT, probs)]
    try:
        L.check(L.lib().a8_set_tuning(14, 0), "tune")
        ref = run()
        L.check(L.lib().a8_set_tuning(14, 1), "tune")
        got = run()
    finally:
        L.check(L.lib().a8_set_tuning(14, 1000), "tune")
    for r, g_ in zip(ref, got):
        assert torch.equal(r, g_)


def test_gemm_tn_asymmetric_ring_matches_symmetric():
    """The asymmetric ring in the weight-gradient layout (a8_set_tuning key 15, default on; both operands K-strided, running source
    pointers): bit-identical to the symmetric ring on integer operands -- plain, split-K with accumulation, ragged M / N / K tails, the
    conv weight gradient's row remaps on both operands (stride-2 input rows, padded gradient rows) and the grouped launch."""
    from audio8_amd import kernels as Kn, _lib as L
    dev, bf = "cuda", torch.bfloat16

    def both(fn):
        try:
            L.check(L.lib().a8_set_tuning(15, 0), "tune")
            ref = fn()
            L.check(L.lib().a8_set_tuning(15, 1), "tune")
            got = fn()
        finally:
            L.check(L.lib().a8_set_tuning(15, 1), "tune")
        return ref, got

    for i, (n_out, n_in, m, sk) in enumerate(((768, 3072, 6248, 1), (3072, 768, 6248, 3), (200, 136, 1000, 1), (128, 128, 64, 1), (512, 1536, 4000, 7))):
        dy, x = _mk((m, n_out), bf, 100 + i, integer=True).to(dev), _mk((m, n_in), bf, 120 + i, integer=True).to(dev)

        def run():
            C = torch.ones(n_out, n_in, device=dev)
            Kn.gemm(L.TN, dy, x, n_out, n_in, m, out=C, accumulate=True, splitk=sk)
            return C
        ref, got = both(run)
        assert torch.equal(ref, got), (n_out, n_in, m, sk)
        np.testing.assert_array_equal(ref.cpu().numpy(), 1.0 + (dy.float().T @ x.float()).cpu().numpy())
    # the conv layers' weight gradient: dW[co][k*cin] = sum over (utterance, frame) of dz[b, pad + t][co] * x[b, s*t + kk][cin]
    Bn, Lin, cin, co, k, s_, pad = 3, 301, 64, 128, 3, 2, 1
    Lout = (Lin - k) // s_ + 1
    xin = _mk((Bn, Lin, cin), bf, 140, integer=True).to(dev)
    dz = _mk((Bn, Lout + 2 * pad, co), bf, 141, integer=True).to(dev)

    def run_conv():
        dw = torch.zeros(co, k * cin, device=dev)
        Kn.gemm(L.TN, dz, xin, co, k * cin, Bn * Lout, out=dw, splitk=3, ldb=s_ * cin, b_rpb=Lout, b_bs=Lin * cin, lda=co,
                a_off=pad * co, a_rpb=Lout, a_bs=(Lout + 2 * pad) * co)
        return dw
    ref, got = both(run_conv)
    assert torch.equal(ref, got)
    cols = torch.stack([xin[:, kk:kk + s_ * (Lout - 1) + 1:s_, :] for kk in range(k)], 2).reshape(Bn * Lout, k * cin).float()
    want = dz[:, pad:pad + Lout, :].reshape(Bn * Lout, co).float().T @ cols
    np.testing.assert_array_equal(ref.cpu().numpy(), want.cpu().numpy())
    # grouped (the encoder layer's four weight gradients)
    M = 1000
    shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
    ops_ = [(_mk((M, a), bf, 150 + j, integer=True).to(dev), _mk((M, b), bf, 160 + j, integer=True).to(dev)) for j, (a, b) in enumerate(shapes)]

    def run_grouped():
        tg = [torch.zeros(a, b, device=dev) for a, b in shapes]
        Kn.gemm_grouped(L.TN, [(dy, x, a, b, M, dict(out=t, accumulate=True)) for (a, b), (dy, x), t in zip(shapes, ops_, tg)])
        return tg
    ref, got = both(run_grouped)
    assert all(torch.equal(a, b) for a, b in zip(ref, got))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_add_epilogue(dtype):
    from audio8_amd import kernels as Kn, _lib as L
    M, N, K = 200, 136, 104
    A, B, R = _mk((M, K), dtype, 31, integer=True), _mk((N, K), dtype, 32, integer=True), _mk((M, N), dtype, 33, integer=True)
    out = Kn.gemm(L.NT, A.cuda(), B.cuda(), M, N, K, act=L.ACT_ADD, zin=R.cuda())
    ref = A.double() @ B.double().T + R.double()
    tol = 0 if dtype == torch.float32 else 1e-2
    np.testing.assert_allclose(out.double().cpu().numpy(), ref.numpy(), rtol=tol, atol=tol)


@pytest.mark.parametrize("key,height", [(5, 0), (5, 160), (5, 192), (5, 224), (5, 256), (14, 0)])
def test_gemm_alternative_nt_kernels_match_default(key, height):
    """The two alternative NT kernels that ship -- the stream kernel of csrc/gemm_stream.hip (a8_set_tuning key 5 = minimum M; N % 256 == 0,
    K % 128 == 0, K >= 256, plain / bias / GELU + pre-activation copy epilogues; BM x 256 tiles on persistent blocks, key 7 forces the tile
    height) and the asymmetric ring (key 14 = 1: A three k-tiles deep) -- walk K in the same order as the default kernel: outputs must be
    bit-identical across M tails, both A-ring depths of the stream kernel (K % 384 == 0 or not), several tiles per persistent block, the
    conv row remaps on A, C and z (batch items that are no multiple of the tile height: their last tile is shifted up), and each
    epilogue; ineligible problems must fall through.  (The round-2 experiments that did not ship live in tools/ubench/gemm_variants.hip.)"""
    from audio8_amd import kernels as Kn, _lib as L
    dev, bf = "cuda", torch.bfloat16
    saved = {5: 0, 14: 0, 7: 0, 19: 0}

    def both(fn):
        try:
            for k_, v_ in saved.items():
                L.check(L.lib().a8_set_tuning(k_, v_), "tune")          # the reference: every alternative kernel off
            ref = fn()
            L.check(L.lib().a8_set_tuning(key, 1), "tune")
            L.check(L.lib().a8_set_tuning(7, height), "tune")
            L.check(L.lib().a8_set_tuning(19, 1), "tune")               # plain (not row-remapped) problems through the stream kernel too
            got = fn()
        finally:
            for k_, v_ in {**saved, 14: 1000, 5: 20000}.items():       # back to the library defaults
                L.check(L.lib().a8_set_tuning(k_, v_), "tune")
        return ref, got

    def same(ref, got, what):
        for r, g_ in zip(ref, got):
            assert torch.equal(r, g_), what

    for (M, N, K) in [(781, 768, 512), (256, 256, 64), (300, 512, 2048), (6248, 768, 768), (1000, 256, 192), (33, 512, 320),
                      (40000, 512, 1536), (129, 1024, 96), (4133, 512, 1024),      # several tiles per persistent block, 2 column tiles
                      (70000, 256, 256), (257, 256, 384), (160, 512, 768),         # 4 k-tiles, more tiles than blocks; one row over a tile; one tile
                      (1000, 264, 72), (300, 136, 2056)]:                  # the last two are not eligible
        A, B = _mk((M, K), bf, 1).to(dev), _mk((N, K), bf, 2).to(dev)
        same(*both(lambda: (Kn.gemm(L.NT, A, B, M, N, K),)), ("plain", M, N, K))
    M, N, K = 6248, 3072, 768
    A, B = _mk((M, K), bf, 3).to(dev), _mk((N, K), bf, 4).to(dev)
    z = _mk((M, N), bf, 5).to(dev)

    def gelu_zout():
        zo = torch.empty(M, N, device=dev, dtype=bf)
        return Kn.gemm(L.NT, A, B, M, N, K, act=L.ACT_GELU, zout=zo, alpha=0.125), zo
    same(*both(gelu_zout), "gelu+zout")
    same(*both(lambda: (Kn.gemm(L.NT, A, B, M, N, K, act=L.ACT_MUL_GELU_GRAD, zin=z),)), "x gelu'(z)")
    same(*both(lambda: (Kn.gemm(L.NT, A, B, M, N, K, act=L.ACT_ADD, zin=z),)), "add")

    # in place (C == zin; M is no multiple of any tile height): the stream kernel's shifted last tile would add zin twice on the rows two
    # tiles store -- stream_plan must hand such a problem to the 128 x 128 kernels (a8hip.h, aliasing rule)
    def add_in_place():
        zc = z.clone()
        return (Kn.gemm(L.NT, A, B, M, N, K, act=L.ACT_ADD, zin=zc, out=zc),)
    ref_ip, got_ip = both(add_in_place)
    same(ref_ip, got_ip, "add, in place")
    assert torch.equal(ref_ip[0], Kn.gemm(L.NT, A, B, M, N, K, act=L.ACT_ADD, zin=z)), "in-place add differs from the out-of-place one"

    def mul_in_place():
        zc = z.clone()
        return (Kn.gemm(L.NT, A, B, M, N, K, act=L.ACT_MUL_GELU_GRAD, zin=zc, out=zc),)
    same(*both(mul_in_place), "x gelu'(z), in place")
    bias = torch.randn(N, device=dev)
    same(*both(lambda: (Kn.gemm(L.NT, A, B, M, N, K, bias=bias),)), "bias")

    def bias_gelu_zout():
        zo = torch.empty(M, N, device=dev, dtype=bf)
        return Kn.gemm(L.NT, A, B, M, N, K, bias=bias, act=L.ACT_GELU, zout=zo), zo
    same(*both(bias_gelu_zout), "bias + gelu + zout")
    same(*both(lambda: (Kn.gemm(L.NT, A, B, M, N, K, out_dtype=torch.float32),)), "f32 out (falls through)")

    def with_colsum():
        cs = torch.zeros(N, device=dev)
        out = Kn.gemm(L.NT, A, B, M, N, K, act=L.ACT_MUL_GELU_GRAD, zin=z, colsum=cs)
        return out, cs
    (r_out, r_cs), (g_out, g_cs) = both(with_colsum)
    assert torch.equal(r_out, g_out)
    torch.testing.assert_close(r_cs, g_cs, rtol=1e-4, atol=1e-2)      # (column sums combine in a different block order)
    # conv forward as a GEMM: overlapping rows of a channels-last activation (k = 3, stride 2), utterance boundaries inside tiles
    Bn, Lin, Cc = 3, 1001, 512
    Lout = (Lin - 3) // 2 + 1
    x = _mk((Bn * Lin * Cc,), bf, 6).to(dev)
    Wc = _mk((512, 3 * Cc), bf, 7).to(dev)

    def conv_fwd():
        zo = torch.empty(Bn * Lout, 512, device=dev, dtype=bf)
        return Kn.gemm(L.NT, x, Wc, Bn * Lout, 512, 3 * Cc, lda=2 * Cc, a_rpb=Lout, a_bs=Lin * Cc, act=L.ACT_GELU, zout=zo), zo
    same(*both(conv_fwd), "conv forward remap")
    # conv data gradient, even positions: rows [dz[t-1] | dz[t]] of a zero-padded dz, output and z rows remapped into dx
    co, cin, npad = 512, 512, 1
    dzrows, prow = Lout + 2, 2 * Lout + 1 + 2 * npad
    dz = _mk((Bn * dzrows, co), bf, 8).to(dev)
    b_even = _mk((cin, 2 * co), bf, 9).to(dev)
    zprev = _mk((Bn * (2 * Lout + 1), cin), bf, 10).to(dev)

    def conv_dgrad_even():
        dx = torch.zeros(Bn * prow, cin, device=dev, dtype=bf)
        Kn.gemm(L.NT, dz, b_even, Bn * (Lout + 1), cin, 2 * co, out=dx, lda=co, a_rpb=Lout + 1, a_bs=dzrows * co,
                ldc=2 * cin, c_off=npad * cin, c_rpb=Lout + 1, c_bs=prow * cin, act=L.ACT_MUL_GELU_GRAD,
                zin=zprev, ldz=2 * cin, z_rpb=Lout + 1, z_bs=(2 * Lout + 1) * cin)
        return (dx,)
    same(*both(conv_dgrad_even), "conv dgrad remap")
    # both halves of the data gradient in one grouped call (odd positions: K = co, output / z rows offset by one frame)
    b_odd = _mk((cin, co), bf, 11).to(dev)

    def conv_dgrad_grouped():
        dx = torch.zeros(Bn * prow, cin, device=dev, dtype=bf)
        zk = dict(zin=zprev, ldz=2 * cin, z_bs=(2 * Lout + 1) * cin)
        even = (dz, b_even, Bn * (Lout + 1), cin, 2 * co,
                dict(out=dx, lda=co, a_rpb=Lout + 1, a_bs=dzrows * co, ldc=2 * cin, c_off=npad * cin, c_rpb=Lout + 1, c_bs=prow * cin,
                     act=L.ACT_MUL_GELU_GRAD, **dict(zk, z_rpb=Lout + 1)))
        odd = (dz, b_odd, Bn * Lout, cin, co,
               dict(out=dx, lda=co, a_off=co, a_rpb=Lout, a_bs=dzrows * co, ldc=2 * cin, c_off=(npad + 1) * cin, c_rpb=Lout, c_bs=prow * cin,
                    act=L.ACT_MUL_GELU_GRAD, **dict(zk, z_rpb=Lout, z_off=cin)))
        Kn.gemm_grouped(L.NT, [even, odd])
        return (dx,)
    same(*both(conv_dgrad_grouped), "conv dgrad, grouped even + odd")




def test_gemm_direct_epilogue_matches_lds_staged():
    """The bf16 NT kernels' LDS-free epilogue (gemm_epilogue_direct: accumulators -> two permlane swaps -> 16-byte stores; no barrier)
    against the LDS-staged one (a8_set_tuning key 22 = 1) on the same main loop: bit-identical outputs over M / N tails, every
    16-bit epilogue (bias, GELU + pre-activation copy, x gelu'(z), + z, in place), alpha != 1, the conv row remaps and the grouped
    launch; problems it does not take (f32 output, column sums) must still run."""
    from audio8_amd import kernels as Kn, _lib as L
    dev, bf = "cuda", torch.bfloat16

    def both(fn):
        try:
            L.check(L.lib().a8_set_tuning(14, 1), "tune")           # every bf16 NT problem on the asymmetric-ring kernel (the one with DIRECT)
            L.check(L.lib().a8_set_tuning(5, 0), "tune")            # stream kernel off
            L.check(L.lib().a8_set_tuning(22, 1), "tune")
            ref = fn()
            L.check(L.lib().a8_set_tuning(22, 0), "tune")
            got = fn()
        finally:
            for k_, v_ in {14: 1000, 5: 20000, 22: 0}.items():
                L.check(L.lib().a8_set_tuning(k_, v_), "tune")
        return ref, got

    def same(pair, what):
        for r, g_ in zip(*pair):
            assert torch.equal(r, g_), what

    for (M, N, K) in [(128, 128, 64), (200, 136, 104), (781, 768, 512), (3124, 768, 768), (130, 264, 72), (33, 8, 64), (1000, 2304, 768)]:
        A, B = _mk((M, K), bf, 1).to(dev), _mk((N, K), bf, 2).to(dev)
        z = _mk((M, N), bf, 3).to(dev)
        bias = torch.randn(N, device=dev)
        same(both(lambda: (Kn.gemm(L.NT, A, B, M, N, K),)), ("plain", M, N, K))
        same(both(lambda: (Kn.gemm(L.NT, A, B, M, N, K, bias=bias, alpha=0.37),)), ("bias, alpha", M, N, K))

        def gelu_zout():
            zo = torch.zeros(M, N, device=dev, dtype=bf)
            return Kn.gemm(L.NT, A, B, M, N, K, bias=bias, act=L.ACT_GELU, zout=zo), zo
        same(both(gelu_zout), ("bias + gelu + zout", M, N, K))
        same(both(lambda: (Kn.gemm(L.NT, A, B, M, N, K, act=L.ACT_MUL_GELU_GRAD, zin=z, alpha=0.5),)), ("x gelu'(z)", M, N, K))
        same(both(lambda: (Kn.gemm(L.NT, A, B, M, N, K, act=L.ACT_ADD, zin=z),)), ("+ z", M, N, K))

        def add_in_place():
            zc = z.clone()
            return (Kn.gemm(L.NT, A, B, M, N, K, act=L.ACT_ADD, zin=zc, out=zc),)
        same(both(add_in_place), ("+ z in place", M, N, K))
        same(both(lambda: (Kn.gemm(L.NT, A, B, M, N, K, out_dtype=torch.float32),)), ("f32 out (LDS path both times)", M, N, K))

        def with_colsum():
            cs = torch.zeros(N, device=dev)
            return Kn.gemm(L.NT, A, B, M, N, K, act=L.ACT_MUL_GELU_GRAD, zin=z, colsum=cs), cs
        (r_out, r_cs), (g_out, g_cs) = both(with_colsum)
        assert torch.equal(r_out, g_out), ("column sums (LDS path both times)", M, N, K)
        torch.testing.assert_close(r_cs, g_cs, rtol=1e-4, atol=1e-2)          # (f32 atomics across blocks: arrival order)
    # reference values, not only self-consistency: exact on small integers
    M, N, K = 300, 136, 96
    A, B, R = _mk((M, K), bf, 31, integer=True), _mk((N, K), bf, 32, integer=True), _mk((M, N), bf, 33, integer=True)
    out = Kn.gemm(L.NT, A.to(dev), B.to(dev), M, N, K, act=L.ACT_ADD, zin=R.to(dev))
    want = (A.double() @ B.double().T + R.double()).to(bf)
    L.check(L.lib().a8_set_tuning(14, 1), "tune")
    try:
        out1 = Kn.gemm(L.NT, A.to(dev), B.to(dev), M, N, K, act=L.ACT_ADD, zin=R.to(dev))
    finally:
        L.check(L.lib().a8_set_tuning(14, 1000), "tune")
    assert torch.equal(out.cpu(), want) and torch.equal(out1.cpu(), want)
    # conv layers as GEMMs: row remaps on A (forward) and on C / z (data gradient, even positions), utterance boundaries inside tiles
    Bn, Lin, Cc = 3, 1001, 512
    Lout = (Lin - 3) // 2 + 1
    x = _mk((Bn * Lin * Cc,), bf, 6).to(dev)
    Wc = _mk((512, 3 * Cc), bf, 7).to(dev)

    def conv_fwd():
        zo = torch.empty(Bn * Lout, 512, device=dev, dtype=bf)
        return Kn.gemm(L.NT, x, Wc, Bn * Lout, 512, 3 * Cc, lda=2 * Cc, a_rpb=Lout, a_bs=Lin * Cc, act=L.ACT_GELU, zout=zo), zo
    same(both(conv_fwd), "conv forward remap")
    co, cin, npad = 512, 512, 1
    dzrows, prow = Lout + 2, 2 * Lout + 1 + 2 * npad
    dz = _mk((Bn * dzrows, co), bf, 8).to(dev)
    b_even, b_odd = _mk((cin, 2 * co), bf, 9).to(dev), _mk((cin, co), bf, 11).to(dev)
    zprev = _mk((Bn * (2 * Lout + 1), cin), bf, 10).to(dev)

    def conv_dgrad_grouped():
        dx = torch.zeros(Bn * prow, cin, device=dev, dtype=bf)
        zk = dict(zin=zprev, ldz=2 * cin, z_bs=(2 * Lout + 1) * cin)
        even = (dz, b_even, Bn * (Lout + 1), cin, 2 * co,
                dict(out=dx, lda=co, a_rpb=Lout + 1, a_bs=dzrows * co, ldc=2 * cin, c_off=npad * cin, c_rpb=Lout + 1, c_bs=prow * cin,
                     act=L.ACT_MUL_GELU_GRAD, **dict(zk, z_rpb=Lout + 1)))
        odd = (dz, b_odd, Bn * Lout, cin, co,
               dict(out=dx, lda=co, a_off=co, a_rpb=Lout, a_bs=dzrows * co, ldc=2 * cin, c_off=(npad + 1) * cin, c_rpb=Lout, c_bs=prow * cin,
                    act=L.ACT_MUL_GELU_GRAD, **dict(zk, z_rpb=Lout, z_off=cin)))
        Kn.gemm_grouped(L.NT, [even, odd])
        return (dx,)
    same(both(conv_dgrad_grouped), "conv dgrad, grouped even + odd")



def test_gemm_lean_addressing_matches_predicate_form_and_k_zero_takes_it():
    """(1) a direct bit-identity A/B of the lean operand addressing of the 128 x 128 NT kernels (KFAST: fixed per-lane
    pointer + block-uniform k offset, clamped tail rows) against the predicate form (a8_set_tuning key 25 = 1) -- whole k-tiles only, M / N
    tails, split-K, the conv row remap, the grouped launch; (2) K = 0 -- an empty reduction, which satisfies K % 64 == 0 -- must NOT take the
    lean form (it would address k-tile -1, 128 bytes in front of every row): C = bias with and without the key, no fault."""
    from audio8_amd import kernels as Kn, _lib as L
    dev, bf = "cuda", torch.bfloat16

    def both(fn):
        try:
            L.check(L.lib().a8_set_tuning(14, 1), "tune")
            L.check(L.lib().a8_set_tuning(5, 0), "tune")
            L.check(L.lib().a8_set_tuning(25, 1), "tune")
            ref = fn()
            L.check(L.lib().a8_set_tuning(25, 0), "tune")
            got = fn()
        finally:
            for k_, v_ in {14: 1000, 5: 20000, 25: 0}.items():
                L.check(L.lib().a8_set_tuning(k_, v_), "tune")
        return ref, got

    for (M, N, K) in [(128, 128, 64), (1000, 768, 768), (3124, 2304, 768), (333, 136, 128), (6248, 768, 3072)]:
        A, B = _mk((M, K), bf, 1).to(dev), _mk((N, K), bf, 2).to(dev)
        z = _mk((M, N), bf, 3).to(dev)
        bias = torch.randn(N, device=dev)
        for what, fn in (("plain", lambda: Kn.gemm(L.NT, A, B, M, N, K)),
                         ("bias + gelu", lambda: Kn.gemm(L.NT, A, B, M, N, K, bias=bias, act=L.ACT_GELU)),
                         ("x gelu'(z)", lambda: Kn.gemm(L.NT, A, B, M, N, K, act=L.ACT_MUL_GELU_GRAD, zin=z)),
                         ("split-K f32", lambda: Kn.gemm(L.NT, A, B, M, N, K, out_dtype=torch.float32, splitk=2 if K >= 128 else 1,
                                                         out=torch.zeros(M, N, device=dev)))):
            r, g_ = both(fn)
            if what == "split-K f32":                       # (atomic partial sums: equal to the order of two f32 adds)
                torch.testing.assert_close(r, g_, rtol=1e-6, atol=1e-6)
            else:
                assert torch.equal(r, g_), (what, M, N, K)
    # a strided conv as a GEMM (overlapping rows): Lin = 100, k = 3, s = 2, cin = 64 -> K = 192
    Bn, Lin, cin, co, kk, st = 3, 101, 64, 128, 3, 2
    Lout = (Lin - kk) // st + 1
    x = _mk((Bn, Lin, cin), bf, 5).to(dev)
    w = _mk((co, kk * cin), bf, 6).to(dev)
    r, g_ = both(lambda: Kn.gemm(L.NT, x, w, Bn * Lout, co, kk * cin, lda=st * cin, a_rpb=Lout, a_bs=Lin * cin))
    assert torch.equal(r, g_), "conv row remap"
    probs = lambda: Kn.gemm_grouped(L.NT, [(_A, _B, m, n, k, {}) for (_A, _B, m, n, k) in grp])       # noqa: E731
    grp = [(_mk((m, k), bf, 7 + i).to(dev), _mk((n, k), bf, 17 + i).to(dev), m, n, k) for i, (m, n, k) in enumerate([(1000, 256, 128), (1300, 128, 256)])]
    r, g_ = both(probs)
    for a_, b_ in zip(r, g_):
        assert torch.equal(a_, b_), "grouped"
    # K = 0
    M, N = 1500, 256
    A0, B0 = torch.empty((M, 0), device=dev, dtype=bf), torch.empty((N, 0), device=dev, dtype=bf)
    bias = torch.randn(N, device=dev)
    for key25 in (0, 1):
        try:
            L.check(L.lib().a8_set_tuning(25, key25), "tune")
            out = Kn.gemm(L.NT, A0, B0, M, N, 0, bias=bias, lda=8, ldb=8)
            torch.cuda.synchronize()
        finally:
            L.check(L.lib().a8_set_tuning(25, 0), "tune")
        assert torch.equal(out, bias.to(bf).expand(M, N)), key25
