#!/bin/bash
# SQ counters of the 128 x 128 (gemm_a3) and the 128 x 256 (gemm_wide) NT kernels on the QKV / FFN1 shapes, alone on the GPU (separate passes)
export TMPDIR=/tmp
out=gpurun_out/pmc_wide; rm -rf $out; mkdir -p $out
for var in "24=0" "24=2304"; do
  i=0
  for set in "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU" "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/v$var/p$i -o c -- python3 tools/enc_gemm_one.py $var > $out/log_$var_$i.txt 2>&1 || tail -3 $out/log_$var_$i.txt
  done
done
python3 - <<'PY' > gpurun_out/pmc_wide/summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_wide/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_" not in k:
            continue
        grid = int(r.get("Grid_Size", 0)) // max(1, int(r.get("Workgroup_Size", 512)))
        name = ("wide 128x256" if "wide" in k else "a3 128x128") + "  blocks %d" % grid
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("SQ / TCP counters per launch, QKV (N 2304) and FFN1 (N 3072, GELU + pre-activation copy) at M = 6248, K = 768, alone on the GPU (rocprofv3 --pmc, separate passes)")
for k in sorted(agg):
    d = agg[k]
    g = {c: sum(v) / len(v) for c, v in d.items()}
    print("\n%s  (%d launches)" % (k, len(next(iter(d.values())))))
    for c in sorted(g):
        print("   %-32s %.4g" % (c, g[c]))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in g and "GRBM_GUI_ACTIVE" in g:
        cyc = g["GRBM_GUI_ACTIVE"] / 8
        print("   -> kernel duration %.0f cycles; MFMA pipe utilisation %.1f %%; waves waiting %.0f %% of their cycles; VALU / MFMA instructions %.2f; SALU / MFMA %.2f" % (
            cyc, 100 * g["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 100 * g.get("SQ_WAIT_ANY", 0) / max(g.get("SQ_WAVE_CYCLES", 1), 1),
            g.get("SQ_INSTS_VALU", 0) / max(g.get("SQ_INSTS_MFMA", 1), 1), g.get("SQ_INSTS_SALU", 0) / max(g.get("SQ_INSTS_MFMA", 1), 1)))
PY
cat gpurun_out/pmc_wide/summary.txt
find gpurun_out/pmc_wide -name "*.csv" -delete
