#!/bin/bash
# HBM bytes per dispatch of the conv-layer-1 GEMM (separate FETCH_SIZE / WRITE_SIZE passes, kernel-trace only)
export TMPDIR=/tmp
out=gpurun_out/conv_probe; rm -rf $out; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -o c -- python3 tools/conv_gemm_probe.py > $out/log_$c.txt 2>&1 || tail -3 $out/log_$c.txt
done
python3 - <<'PY'
import csv, glob
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/conv_probe/%s/**/*counter_collection.csv" % c, recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "gemm_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c]
    print(c, "per dispatch (KiB; FETCH_SIZE x2 on gfx950):", [round(v / 1024, 1) for v in vals], "MiB")
PY
grep variant $out/log_FETCH_SIZE.txt
find $out -name "*.csv" -delete
