#!/bin/bash
# kernel-trace profile of the bench step with the weight-gradient side stream and the half-batch streams off: every kernel's
# own (un-overlapped) cost
export TMPDIR=/tmp A8_SIDE_WGRAD=0 A8_HALF_BATCH=0
mkdir -p gpurun_out/prof_serial
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_serial -o s -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --h2d-steps 0 > gpurun_out/prof_serial/log.txt 2>&1
python3 tools/prof_summary.py gpurun_out/prof_serial 7 45 > gpurun_out/prof_serial/summary.txt 2>&1
python3 tools/trace_timeline.py gpurun_out/prof_serial 7 >> gpurun_out/prof_serial/summary.txt 2>&1
rm -f gpurun_out/prof_serial/*trace.csv gpurun_out/prof_serial/*/*trace.csv
tail -3 gpurun_out/prof_serial/log.txt
