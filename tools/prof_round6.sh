#!/bin/bash
# Round-6 profile artifacts in one GPU call: kernel-trace stats of the default bench command, the FETCH_SIZE / WRITE_SIZE PMC
# passes (own runs, kernel-trace only) for the whole step and for the conv-stack forward alone, SQ counters of the GEMM kernels.
# Results land in gpurun_out/prof_r06/ (copy what is to be judged into profiles/).
export TMPDIR=/tmp
out=gpurun_out/prof_r06
rm -rf $out; mkdir -p $out
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --h2d-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o s -- $CMD > $out/trace_log.txt 2>&1 || tail -5 $out/trace_log.txt
python3 tools/prof_summary.py $out/trace 6 50 > $out/summary.txt 2>&1
python3 tools/trace_timeline.py $out/trace 6 >> $out/summary.txt 2>&1
cp $out/trace/s_kernel_stats.csv $out/kernel_stats.csv
rm -rf $out/trace
echo "trace done"; head -3 $out/summary.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o c -- $CMD > $out/fetch_log.txt 2>&1 || tail -5 $out/fetch_log.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o c -- $CMD > $out/write_log.txt 2>&1 || tail -5 $out/write_log.txt
echo "step pmc done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/cfetch -o c -- python3 tools/conv_only.py 3 > $out/cfetch_log.txt 2>&1 || tail -5 $out/cfetch_log.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/cwrite -o c -- python3 tools/conv_only.py 3 > $out/cwrite_log.txt 2>&1 || tail -5 $out/cwrite_log.txt
python3 tools/pmc_families.py $out/fetch $out/write 6 $out/gemm_traffic.json $out/cfetch $out/cwrite 3 > $out/traffic_log.txt 2>&1 || tail -5 $out/traffic_log.txt
rm -rf $out/fetch $out/write $out/cfetch $out/cwrite
echo "conv pmc done"; tail -25 $out/traffic_log.txt
bash tools/pmc_step_kernels.sh > $out/pmc_step_log.txt 2>&1; cp gpurun_out/pmc_step/summary.txt $out/kernel_sq_counters.txt
bash tools/prof_serial.sh > $out/serial_log.txt 2>&1; cp gpurun_out/prof_serial/summary.txt $out/serial_summary.txt 2>/dev/null
echo "sq done"
