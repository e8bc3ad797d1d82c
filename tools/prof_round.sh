#!/bin/bash
# Everything the round's profile artifacts need, in one GPU call: kernel-trace stats of the default bench command and
# the two PMC passes (FETCH_SIZE, WRITE_SIZE; own runs, kernel-trace only).  Results land in gpurun_out/prof_round/.
export TMPDIR=/tmp
out=gpurun_out/prof_round
rm -rf $out; mkdir -p $out
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o s -- $CMD > $out/trace_log.txt 2>&1 || tail -5 $out/trace_log.txt
python3 tools/prof_summary.py $out/trace 6 45 > $out/summary.txt 2>&1
python3 tools/trace_timeline.py $out/trace 6 >> $out/summary.txt 2>&1
cp $out/trace/s_kernel_stats.csv $out/kernel_stats.csv
rm -f $out/trace/*trace.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o c -- $CMD > $out/fetch_log.txt 2>&1 || tail -5 $out/fetch_log.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o c -- $CMD > $out/write_log.txt 2>&1 || tail -5 $out/write_log.txt
mkdir -p $out/fetch/x $out/write/x; mv $out/fetch/*counter_collection.csv $out/fetch/x/ 2>/dev/null; mv $out/write/*counter_collection.csv $out/write/x/ 2>/dev/null
python3 tools/pmc_traffic.py $out/fetch $out/write $out/gemm_traffic.json 6 > $out/traffic_log.txt 2>&1 || tail -5 $out/traffic_log.txt
rm -rf $out/fetch $out/write $out/trace
tail -3 $out/traffic_log.txt; head -3 $out/summary.txt
