#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/cpn; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -o s -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/log.txt 2>&1
python3 - <<'PY'
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("gpurun_out/cpn/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
step = rows[ends[-2] + 1: ends[-1] + 1]
def short(n):
    import re
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    return n[:48]
c = collections.Counter()
for i, r in enumerate(step):
    if "copyBuffer" in r["Kernel_Name"]:
        prev = short(step[i - 1]["Kernel_Name"]) if i else "-"
        nxt = short(step[i + 1]["Kernel_Name"]) if i + 1 < len(step) else "-"
        c[(prev, nxt)] += 1
for k, v in c.most_common(30):
    print(v, k)
PY
rm -f $out/*trace.csv
