#!/usr/bin/env python3
"""Gate of VERDICT r5 item 1 ("one persistent launch per encoder-layer GEMM set"): the eight GEMMs of one encoder layer's BACKWARD at the cfg2
shape (M = 8 x 781 = 6248 rows, d = 768, d_ff = 3072), alone on the GPU, operands cold (three rotating operand sets of ~230 MB each):

  A  today's arrangement: the four data gradients as two half-batch chains (128 x 128-tile gemm_a3 kernels) on two streams, the four weight
     gradients as one grouped TN launch on a third stream -- all three released together;
  B  the arrangement the item asks for, with the kernels that exist: whole batch, ONE stream, the four data gradients on the persistent
     BM x 256-tile stream kernel (a8_set_tuning 19 = 1, 5 = 1000), then the grouped TN launch -- nothing shares a CU;
  C  a bound for a work-list kernel that packs all eight products' tiles perfectly: their FLOPs at the best steady rate each kernel family has
     ever reached on this part (NT 1.36 PFLOP/s: profiles/r04_stream_gemm_standalone.txt; TN 0.75: DESIGN section 5).

The products are launched WITHOUT the LayerNorm / attention kernels that separate them in the real chain (FFN2 dgrad -> FFN1 dgrad -> LN ->
w_O dgrad -> attention -> QKV dgrad): B and C therefore overstate what a single launch could gain -- it would have to wait at those points.
Integrate only if B is >= 25 % faster than A."""
import os
import sys

os.sched_setaffinity(0, sorted(os.sched_getaffinity(0))[:4])
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from audio8_amd import _lib as L, kernels as K  # noqa: E402

M, D, F = 6248, 768, 3072
bf = torch.bfloat16
dev = "cuda"


def tune(k, v):
    L.check(L.lib().a8_set_tuning(k, v), "tune")


def make_set(seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    r = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.05).to(bf)      # noqa: E731
    return dict(df=r(M, D), z1=r(M, F), dz1=torch.empty(M, F, device=dev, dtype=bf), dx1=torch.empty(M, D, device=dev, dtype=bf), dao=r(M, D),
                do=torch.empty(M, D, device=dev, dtype=bf), dqkv=r(M, 3 * D), dxres=r(M, D), dx=torch.empty(M, D, device=dev, dtype=bf),
                h=r(M, F), x1=r(M, D), o=r(M, D), x=r(M, D),
                w2t=r(F, D), w1t=r(D, F), wot=r(D, D), wqkvt=r(D, 3 * D),
                gw2=torch.zeros(D, F, device=dev), gw1=torch.zeros(F, D, device=dev), gwo=torch.zeros(D, D, device=dev), gwqkv=torch.zeros(3 * D, D, device=dev))


def dgrads(s, r0, r1):
    m = r1 - r0
    K.gemm(L.NT, s["df"][r0:r1], s["w2t"], m, F, D, act=L.ACT_MUL_GELU_GRAD, zin=s["z1"][r0:r1], ldz=F, out=s["dz1"][r0:r1])
    K.gemm(L.NT, s["dz1"][r0:r1], s["w1t"], m, D, F, out=s["dx1"][r0:r1])
    K.gemm(L.NT, s["dao"][r0:r1], s["wot"], m, D, D, out=s["do"][r0:r1])
    K.gemm(L.NT, s["dqkv"][r0:r1], s["wqkvt"], m, D, 3 * D, act=L.ACT_ADD, zin=s["dxres"][r0:r1], ldz=D, out=s["dx"][r0:r1])


def wgrads(s):
    K.gemm_grouped(L.TN, [(s["df"], s["h"], D, F, M, dict(out=s["gw2"], accumulate=True)),
                          (s["dz1"], s["x1"], F, D, M, dict(out=s["gw1"], accumulate=True)),
                          (s["dao"], s["o"], D, D, M, dict(out=s["gwo"], accumulate=True)),
                          (s["dqkv"], s["x"], 3 * D, D, M, dict(out=s["gwqkv"], accumulate=True))])


def main():
    sets = [make_set(i) for i in range(3)]
    s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    half = (8 // 2) * 781

    def arrangement_a(s):
        cur = torch.cuda.current_stream()
        for st in (s1, s2, s3):
            st.wait_stream(cur)
        with torch.cuda.stream(s1):
            dgrads(s, 0, half)
        with torch.cuda.stream(s2):
            dgrads(s, half, M)
        with torch.cuda.stream(s3):
            wgrads(s)
        for st in (s1, s2, s3):
            cur.wait_stream(st)

    def arrangement_b(s):
        dgrads(s, 0, M)
        wgrads(s)

    def timed(fn, n=30):
        for i in range(3):
            fn(sets[i % 3])
        torch.cuda.synchronize()
        evs = []
        for i in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(sets[i % 3])
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
        return t[len(t) // 2], t[0]

    flop_nt = 2.0 * M * (F * D + D * F + D * D + D * 3 * D)
    flop_tn = flop_nt
    print("# tools/ubench/layer_set_ab.py: one encoder layer's backward GEMM set (4 data gradients + 4 weight gradients, %.1f GFLOP), cfg2 shape, cold operands" % ((flop_nt + flop_tn) / 1e9))
    for rnd in range(3):
        tune(19, 0); tune(5, 20000)
        a_med, a_min = timed(arrangement_a)
        tune(19, 1); tune(5, 1000)
        b_med, b_min = timed(arrangement_b)
        tune(19, 0); tune(5, 20000)
        b2_med, b2_min = timed(arrangement_b)
        print("round %d: A three streams, 128 x 128 tiles %.1f us (min %.1f) | B one stream, persistent BM x 256 data gradients %.1f us (min %.1f) -> B / A = %.3f | "
              "(one stream, 128 x 128 tiles: %.1f us)" % (rnd, a_med, a_min, b_med, b_min, b_med / a_med, b2_med))
    bound = (flop_nt / 1.36e15 + flop_tn / 0.75e15) * 1e6
    print("C perfect packing at the families' best steady rates (NT 1.36, TN 0.75 PFLOP/s): %.1f us" % bound)


if __name__ == "__main__":
    main()
