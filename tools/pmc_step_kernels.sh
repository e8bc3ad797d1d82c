#!/bin/bash
# SQ counters, per kernel family, of the bench step as it runs (rocprofv3 --kernel-trace --pmc, separate passes): MFMA / VALU / LDS shares,
# instruction classes (transcendental, convert, integer), waits -- VERDICT r4 items 1, 2, 6 ask for these by name.
export TMPDIR=/tmp
out=gpurun_out/pmc_step; rm -rf $out; mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --h2d-steps 0 > $out/log$i.txt 2>&1 || tail -3 $out/log$i.txt
done
python3 - <<'PY' > gpurun_out/pmc_step/summary.txt
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
FAM = [("attn_bwd_kv", "attn_bwd_kv"), ("attn_bwd_q", "attn_bwd_q"), ("attn_fwd32", "attn_fwd32"), ("conv0_bwd_mfma", "conv0_bwd_mfma"), ("conv0_apply_mfma", "conv0_apply_mfma"),
       ("ln_bwd", "ln_bwd_kernel"), ("ln_fwd", "ln_fwd_kernel"), ("gemm_a3 (NT 128x128)", "gemm_a3_kernel"), ("gemm_grouped_a3_tn (encoder wgrad)", "gemm_grouped_a3_tn"),
       ("gemm_a3_tn (conv wgrad)", "gemm_a3_tn_kernel"), ("gemm_stream (conv NT)", "gemm_stream"), ("adamw", "adamw_kernel"), ("colsum_partials_reduce", "colsum_partials"),
       ("gelu_bwd_bias", "gelu_bwd_bias"), ("posconv_fwd", "posconv_fwd"), ("posconv_wgrad", "posconv_wgrad")]
for f in glob.glob("gpurun_out/pmc_step/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "IfE" in k or "<float" in k:
            continue
        for name, pat in FAM:
            if pat in k:
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
                break
print("SQ counters per launch, by kernel family, in the bench step as it runs (bench.py --steps 2 --warmup 1; rocprofv3 --kernel-trace --pmc, 4 separate passes)")
print("SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE summed over 8 XCDs")
for name, _ in FAM:
    d = agg.get(name)
    if not d:
        continue
    g = {c: sum(v) / len(v) for c, v in d.items()}
    print("\n%s  (%d launches)" % (name, len(next(iter(d.values())))))
    for c in sorted(g):
        print("   %-32s %.4g" % (c, g[c]))
    cyc = g.get("GRBM_GUI_ACTIVE", 0) / 8
    mf, va = g.get("SQ_INSTS_MFMA", 0), g.get("SQ_INSTS_VALU", 0)
    line = "   -> duration %.0f cycles" % cyc
    if cyc and "SQ_VALU_MFMA_BUSY_CYCLES" in g:
        line += "; MFMA pipe %.1f %% busy" % (100 * g["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc))
    if "SQ_WAIT_ANY" in g and g.get("SQ_WAVE_CYCLES"):
        line += "; waves waiting %.0f %%, issue-stalled %.0f %%, issuing %.0f %% of their cycles" % (100 * g["SQ_WAIT_ANY"] / g["SQ_WAVE_CYCLES"], 100 * g.get("SQ_WAIT_INST_ANY", 0) / g["SQ_WAVE_CYCLES"], 100 * g.get("SQ_ACTIVE_INST_ANY", 0) / g["SQ_WAVE_CYCLES"])
    if mf:
        line += "; per MFMA: %.2f VALU (%.2f transcendental, %.2f convert, %.2f integer), %.2f SALU, %.2f LDS" % (
            (va - mf) / mf if va > mf else va / mf, g.get("SQ_INSTS_VALU_TRANS_F32", 0) / mf, g.get("SQ_INSTS_VALU_CVT", 0) / mf, g.get("SQ_INSTS_VALU_INT32", 0) / mf,
            g.get("SQ_INSTS_SALU", 0) / mf, g.get("SQ_INSTS_LDS", 0) / mf)
    print(line)
PY
cat gpurun_out/pmc_step/summary.txt
find gpurun_out/pmc_step -name "*.csv" -delete
