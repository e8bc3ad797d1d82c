#!/bin/bash
# SQ counters of the GEMM kernels in the bench step (separate passes, kernel-trace only): MFMA pipe utilisation, LDS, waits
export TMPDIR=/tmp
out=gpurun_out/pmc_gemm; rm -rf $out; mkdir -p $out
i=0
for set in "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/log$i.txt 2>&1 || tail -3 $out/log$i.txt
done
python3 - <<'PY' > gpurun_out/pmc_gemm/summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_gemm/p*/**/*counter_collection.csv", recursive=True) + glob.glob("gpurun_out/pmc_gemm/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_" not in k or "<float" in k or "IfL" in k:
            continue
        name = "gemm_stream(NT,bf16,BMx256)" if "gemm_stream" in k else "gemm_grouped(TN,bf16)" if "grouped" in k else ("gemm(TN,bf16)" if ("Lb1ELb1" in k or "true, true" in k or "E, true" in k) else "gemm(NT,bf16)")
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("SQ counters of the GEMM kernels during `bench.py --steps 2 --warmup 1` (rocprofv3 --pmc, 4 separate passes); averages per launch")
for k, d in agg.items():
    n = len(next(iter(d.values())))
    print("\n%s  (%d launches)" % (k, n))
    g = {c: sum(v) / len(v) for c, v in d.items()}
    for c in sorted(g):
        print("   %-32s %.4g" % (c, g[c]))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in g and "GRBM_GUI_ACTIVE" in g:
        # MFMA busy cycles are summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        cyc = g["GRBM_GUI_ACTIVE"] / 8
        print("   -> kernel duration %.0f cycles; MFMA pipe utilisation = MFMA_BUSY / (1024 SIMDs x cycles) = %.1f %%" % (cyc, 100 * g["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc)))
PY
cat gpurun_out/pmc_gemm/summary.txt
find gpurun_out/pmc_gemm -name "*.csv" -delete
