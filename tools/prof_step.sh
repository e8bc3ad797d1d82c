#!/bin/bash
# kernel-trace profile of the default bench step (side stream on): per-kernel summary + queue timeline
export TMPDIR=/tmp
out=gpurun_out/prof_step
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $out/log.txt 2>&1
python3 tools/prof_summary.py $out 7 45 > $out/summary.txt 2>&1
python3 tools/trace_timeline.py $out 7 $1 >> $out/summary.txt 2>&1
cp $out/s_kernel_stats.csv $out/kernel_stats.csv 2>/dev/null
rm -f $out/*trace.csv
tail -2 $out/log.txt | cut -c1-200
