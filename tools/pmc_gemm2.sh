#!/bin/bash
# Memory-path counters of three representative bf16 NT GEMMs launched alone (tools/gemm_one.py: FFN1-like 6248 x 3072 x 768,
# O-projection-like 6248 x 768 x 768, conv-layer-1-like 200000 x 512 x 1536): average outstanding-instruction levels
# (level / instructions = average latency in cycles)
# NOTE: a pass with TA_* counters (TA_BUSY, TA_ADDR_STALLED_BY_TC_CYCLES ...) aborts rocprofv3 on this image (signal 6, then a hang): left out., FIFO-full stalls, L1 -> L2 request latency, L2 hit rate, L2 -> fabric read level.
export TMPDIR=/tmp
out=gpurun_out/pmc_gemm2; rm -rf $out; mkdir -p $out
i=0
for set in "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL" "TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES" "TCC_HIT TCC_MISS TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "TCP_TOTAL_READ TCP_TCP_TA_DATA_STALL_CYCLES TCP_TD_TCP_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o c -- python3 tools/gemm_one.py > $out/log$i.txt 2>&1 || tail -3 $out/log$i.txt
done
python3 - <<'PY' > gpurun_out/pmc_gemm2/summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_gemm2/p*/**/*counter_collection.csv", recursive=True) + glob.glob("gpurun_out/pmc_gemm2/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_" not in k:
            continue
        name = {2352: "ffn1-like (1176 tiles)", 588: "o-like (294 tiles)", 12504: "conv1-like (6252 tiles)"}.get(int(r["Grid_Size"]) // 256, "grid %s" % r["Grid_Size"])
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("rocprofv3 --pmc on tools/gemm_one.py (each GEMM alone, 3 launches, averages per launch; counters are summed over the chip)")
for k, d in sorted(agg.items()):
    g = {c: sum(v) / len(v) for c, v in d.items()}
    print("\n%s" % k)
    for c in sorted(g):
        print("   %-36s %.4g" % (c, g[c]))
    def ratio(a, b, what):
        if a in g and b in g and g[b]:
            print("   -> %s = %.1f" % (what, g[a] / g[b]))
    ratio("SQ_INST_LEVEL_VMEM", "SQ_INSTS_VMEM_RD", "average VMEM (LDS-DMA) latency, cycles (level / read instructions; stores included in the level)")
    ratio("SQ_INST_LEVEL_LDS", "SQ_INSTS_LDS", "average LDS instruction latency, cycles")
    ratio("TCP_TCC_READ_REQ_LATENCY", "TCP_TCC_READ_REQ", "average L1->L2 read request latency, cycles")
    ratio("TCC_EA0_RDREQ_LEVEL", "TCC_EA0_RDREQ", "average L2->fabric read latency, cycles")
    if "TCC_HIT" in g:
        print("   -> L2 hit rate = %.1f %%" % (100 * g["TCC_HIT"] / (g["TCC_HIT"] + g["TCC_MISS"])))
    ratio("SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "fraction of wave-cycles waiting")
PY
cat gpurun_out/pmc_gemm2/summary.txt
find gpurun_out/pmc_gemm2 -name "*.csv" -delete
