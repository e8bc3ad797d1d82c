"""a8_adamw took 516 us in round 2 and 674-729 us in round 4 on the same 3.04 GB.  This probe runs the kernel
alone on flat buffers of the base model's size and reports the distribution of its duration (HIP events), the clocks rocm-smi shows,
and how the time moves with (a) the launch grid, (b) nontemporal loads / stores, (c) what ran just before it (cold: after a 1 GB fill of
another buffer; warm: back to back), (d) one launch vs the per-run split (ranges of the flat buffers), (e) zero_grad on / off.
Usage: python tools/adamw_probe.py [reps] > profiles/r05_adamw.txt"""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio8_amd import kernels as K, _lib as L      # noqa: E402


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showperflevel"], capture_output=True, text=True, timeout=30).stdout
        keep = [l.strip() for l in out.splitlines() if any(t in l for t in ("sclk", "mclk", "fclk", "Power", "Performance Level"))]
        return "; ".join(keep[:8])
    except Exception as e:      # noqa: BLE001
        return f"rocm-smi unavailable: {e}"


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    n = 95_044_000 // 4 * 4
    dev = "cuda"
    p, m, v = (torch.randn(n, device=dev) * 0.02 for _ in range(3))
    v.abs_()
    g = torch.randn(n, device=dev) * 1e-3
    other = torch.empty(256 * 1024 * 1024, device=dev)          # 1 GB: evicts the Infinity Cache between two launches
    gn = torch.ones(1, device=dev)
    print(f"a8_adamw alone, n = {n} f32 elements x 4 buffers (read 16 B + write 16 B per element with zero_grad = {32 * n / 1e9:.2f} GB)")
    print("rocm-smi before:", smi())

    def run(tag, grid=4096, nt=4, cold=False, zero=True, pieces=1, reps_=reps):
        L.check(L.lib().a8_set_tuning(20, grid), "tune")
        L.check(L.lib().a8_set_tuning(21, nt), "tune")
        ts = []
        bounds = [n * i // pieces // 4 * 4 for i in range(pieces)] + [n]
        for r in range(reps_ + 3):
            if cold:
                other.fill_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(pieces):
                lo, hi = bounds[i], bounds[i + 1]
                K.adamw_(p[lo:hi], g[lo:hi], m[lo:hi], v[lo:hi], 1e-4, 0.9, 0.98, 1e-6, 0.01, r + 1, gn, 1.0, 1.0, zero_grad=zero)
            e1.record()
            torch.cuda.synchronize()
            if r >= 3:
                ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        byt = (32 if zero else 28) * n
        med = ts[len(ts) // 2]
        print(f"{tag:58s} min {ts[0]:6.1f}  p10 {ts[len(ts) // 10]:6.1f}  median {med:6.1f}  p90 {ts[9 * len(ts) // 10]:6.1f}  max {ts[-1]:6.1f} us"
              f"   median {byt / med / 1e6:5.2f} TB/s")
        return med

    run("shipped default (grid 65536, nontemporal loads + stores)", grid=0, nt=0)
    run("shipped default, cold", grid=0, nt=0, cold=True)
    run("round 2-4 geometry (grid 4096 x 256, plain loads/stores), back to back")
    run("round 2-4 geometry, after a 1 GB fill of another buffer (cold)", cold=True)
    for grid in (1024, 2048, 8192, 16384, 32768, 65536, 1 << 20):
        run(f"grid {grid}", grid=grid)
    for grid in (16384, 65536, 1 << 20):
        run(f"grid {grid}, nontemporal loads + stores", grid=grid, nt=3)
        run(f"grid {grid}, cold", grid=grid, cold=True)
    for nt, name in ((1, "nontemporal loads"), (2, "nontemporal stores"), (3, "nontemporal loads + stores")):
        run(name, nt=nt)
        run(name + ", cold", nt=nt, cold=True)
    run("zero_grad off (28 B per element)", zero=False)
    run("split into 4 launches (ranges of the flat buffers)", pieces=4)
    run("split into 40 launches", pieces=40)
    run("round 2-4 geometry again (drift check)")
    print("rocm-smi after:", smi())
    L.check(L.lib().a8_set_tuning(20, 0), "tune")
    L.check(L.lib().a8_set_tuning(21, 0), "tune")


if __name__ == "__main__":
    main()
