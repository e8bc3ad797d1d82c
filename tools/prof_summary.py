#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-kernel ms/step and per-shape GEMM timings."""
import collections
import csv
import glob
import re
import sys


def main():
    d, nsteps = sys.argv[1], int(sys.argv[2])
    st = glob.glob(d + "/*kernel_stats.csv")[0]
    tr = glob.glob(d + "/*kernel_trace.csv")[0]
    rows = list(csv.DictReader(open(st)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"total kernel time {tot / 1e6 / nsteps:.3f} ms/step over {nsteps} steps")
    for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
        n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)[:70]
        print(f"{float(r['TotalDurationNs']) / 1e6 / nsteps:8.3f} ms/step  calls/step {int(r['Calls']) / nsteps:6.1f}  avg {float(r['AverageNs']) / 1e3:8.1f} us  {float(r['Percentage']):5.1f}%  {n}")
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
