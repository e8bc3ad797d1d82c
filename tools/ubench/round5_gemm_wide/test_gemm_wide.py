"""The bit-identity test the experiment passed (57 / 57 green on MI355X with the kernel wired into a8_gemm behind a8_set_tuning key 24); kept with
the kernel, outside the product test-suite.  Needs helpers of tests/test_gemm_gpu.py (_mk)."""
import pytest
import torch
from test_gemm_gpu import _mk

pytestmark = pytest.mark.gpu


def test_gemm_wide_128x256_kernel_matches_default():
    """csrc/gemm_wide.hip (128 x 256 tiles, 32-deep k-tiles in a 4 + 3 stage ring, two blocks per CU; a8_set_tuning key 24 = minimum N) against
    the 128 x 128 kernels: bit-identical 16-bit outputs over M tails (clamped source rows), K from three k-tiles up, several column tiles and
    every epilogue it takes (bias, GELU + pre-activation copy, x gelu'(z), + z, in place, alpha); column sums to atomic-order tolerance;
    problems it must refuse (N % 256 != 0, f32 output, row remaps, K < 96) still run."""
    from audio8_amd import kernels as Kn, _lib as L
    dev, bf = "cuda", torch.bfloat16

    def both(fn):
        try:
            L.check(L.lib().a8_set_tuning(24, 0), "tune")
            L.check(L.lib().a8_set_tuning(5, 0), "tune")
            ref = fn()
            L.check(L.lib().a8_set_tuning(24, 256), "tune")
            got = fn()
        finally:
            for k_, v_ in {24: 2304, 5: 20000}.items():
                L.check(L.lib().a8_set_tuning(k_, v_), "tune")
        return ref, got

    def same(pair, what):
        for r, g_ in zip(*pair):
            assert torch.equal(r, g_), what

    for (M, N, K) in [(128, 256, 96), (200, 512, 128), (781, 768, 512), (3124, 2304, 768), (6248, 3072, 768), (130, 256, 2048), (1, 256, 160),
                      (3124, 768, 3072), (257, 1024, 96), (300, 264, 128), (300, 256, 64)]:        # the last two are refused (N % 256, K < 96)
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
        same(both(lambda: (Kn.gemm(L.NT, A, B, M, N, K, out_dtype=torch.float32),)), ("f32 out (refused)", M, N, K))

        def with_colsum():
            cs = torch.zeros(N, device=dev)
            return Kn.gemm(L.NT, A, B, M, N, K, act=L.ACT_MUL_GELU_GRAD, zin=z, colsum=cs), cs
        (r_out, r_cs), (g_out, g_cs) = both(with_colsum)
        assert torch.equal(r_out, g_out), ("column sums: outputs", M, N, K)
        torch.testing.assert_close(r_cs, g_cs, rtol=1e-4, atol=2e-2 * (M ** 0.5))          # (f32 atomics: arrival order; per-wave instead of per-block sums)
    # exact reference on small integers (not only self-consistency), an M tail inside the last tile
    M, N, K = 300, 512, 160
    A, B, R = _mk((M, K), bf, 31, integer=True), _mk((N, K), bf, 32, integer=True), _mk((M, N), bf, 33, integer=True)
    L.check(L.lib().a8_set_tuning(24, 256), "tune")
    try:
        out = Kn.gemm(L.NT, A.to(dev), B.to(dev), M, N, K, act=L.ACT_ADD, zin=R.to(dev))
    finally:
        L.check(L.lib().a8_set_tuning(24, 2304), "tune")
    assert torch.equal(out.cpu(), (A.double() @ B.double().T + R.double()).to(bf))
    # a row-remapped problem (conv layer as a GEMM) must be refused and still come out right
    Bn, Lin, Cc = 2, 401, 512
    Lout = (Lin - 3) // 2 + 1
    x = _mk((Bn * Lin * Cc,), bf, 6).to(dev)
    Wc = _mk((512, 3 * Cc), bf, 7).to(dev)
    same(both(lambda: (Kn.gemm(L.NT, x, Wc, Bn * Lout, 512, 3 * Cc, lda=2 * Cc, a_rpb=Lout, a_bs=Lin * Cc),)), "conv remap (refused)")
