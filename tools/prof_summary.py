#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-kernel ms/step and per-shape GEMM timings."""
import collections
import csv
import glob
import re
import sys


def main():
    d, nsteps = sys.argv[1], int(sys.argv[2])
    tr = glob.glob(d + "/*kernel_trace.csv")[0]
    # per-kernel table from the TRACE, restricted to the steady steps (after the first optimizer step: the --stats CSV also
    # holds model set-up copies and every kernel's first launch, which includes loading its code object)
    trace = list(csv.DictReader(open(tr)))
    trace.sort(key=lambda r: int(r["Start_Timestamp"]))
    # one optimizer step = one sumsq_final launch (AdamW itself is one launch per run of parameters that have a gradient: two or more
    # per step with a frozen feature extractor, cfg5); traces without the fused norm fall back to the AdamW launches
    ends = [i for i, r in enumerate(trace) if "sumsq_final" in r["Kernel_Name"]] or [i for i, r in enumerate(trace) if "adamw" in r["Kernel_Name"]]
    steady, nst = trace[ends[0] + 1: ends[-1] + 1], len(ends) - 1
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in steady:
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot = sum(a[1] for a in agg.values())
    print(f"total kernel time {tot / 1e6 / nst:.3f} ms/step over the {nst} steady steps of the trace (of {nsteps} run; the last two are bench.py's instrumented "
          f"roofline steps, the very last one with the weight-gradient side stream off)")
    top = sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]
    for name, (calls, ns) in top:
        n = re.sub(r"\(anonymous namespace\)::", "", name)
        n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)[:70]
        print(f"{ns / 1e6 / nst:8.3f} ms/step  calls/step {calls / nst:6.1f}  avg {ns / calls / 1e3:8.1f} us  {100 * ns / tot:5.1f}%  {n}")
    g = [r for r in csv.DictReader(open(tr)) if "gemm_kernel" in r["Kernel_Name"]]
    n = len(g) // nsteps
    agg = collections.OrderedDict()
    for r in g[-n:]:
        nm = r["Kernel_Name"]
        kind = "TN" if ("Lb1ELb1" in nm or "true, true" in nm or "E, true" in nm) else ("NN" if ("Lb0ELb1" in nm or "false, true" in nm) else "NT")
        if "IfL" in nm or "float" in nm:
            kind = "f32-" + kind
        agg.setdefault((kind, int(r["Grid_Size_X"]) // 256), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("GEMM dispatches of the last step (layout, blocks): n, avg us, total ms")
    for k, v in agg.items():
        print(f"  {k}  n {len(v):3d}  avg {sum(v) / len(v):8.1f} us  total {sum(v) / 1e3:6.2f} ms")


if __name__ == "__main__":
    main()
