#!/bin/bash
# Round 6, second pass (after the posconv_wgrad and CTC kernel changes): the artifacts those kernels appear in, refreshed in one GPU call --
# kernel-trace stats of the default bench command, the serialised step, the SQ counter passes, cfg5 / cfg4 traces and their ms/step lines.
# Results land in gpurun_out/prof_r06b/ (copy what is to be judged into profiles/).
export TMPDIR=/tmp
out=gpurun_out/prof_r06b
rm -rf $out; mkdir -p $out
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --h2d-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o s -- $CMD > $out/trace_log.txt 2>&1 || tail -5 $out/trace_log.txt
python3 tools/prof_summary.py $out/trace 6 50 > $out/summary.txt 2>&1
python3 tools/trace_timeline.py $out/trace 6 >> $out/summary.txt 2>&1
cp $out/trace/s_kernel_stats.csv $out/kernel_stats.csv
rm -rf $out/trace
echo "trace done"; head -3 $out/summary.txt
bash tools/prof_serial.sh > $out/serial_log.txt 2>&1; cp gpurun_out/prof_serial/summary.txt $out/serial_summary.txt 2>/dev/null
echo "serial done"
bash tools/prof_configs.sh $out/cfg > $out/cfg_log.txt 2>&1
echo "configs done"
python3 tools/run_configs.py both 20 > $out/configs.txt 2>&1; grep "^cfg" $out/configs.txt
bash tools/pmc_step_kernels.sh > $out/pmc_step_log.txt 2>&1; cp gpurun_out/pmc_step/summary.txt $out/kernel_sq_counters.txt
echo "sq done"
