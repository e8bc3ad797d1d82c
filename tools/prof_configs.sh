#!/bin/bash
# Kernel traces of BASELINE.json configs 5 (CTC fine-tune, 16 x 200k) and 4 (wav2vec2-large, 4 x 320k) at full size:
# rocprofv3 --kernel-trace --stats of tools/run_configs.py (2 warm-up + 3 timed steps each), summarised per kernel over the steady steps.
# usage: bash tools/prof_configs.sh <outdir>     (copy the summaries to be judged into profiles/)
export TMPDIR=/tmp
out=${1:-gpurun_out/prof_cfg}
mkdir -p $out
for c in cfg5 cfg4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$c -o s -- python3 tools/run_configs.py $c 3 > $out/${c}_log.txt 2>&1 || tail -5 $out/${c}_log.txt
  python3 tools/prof_summary.py $out/trace_$c 5 60 > $out/${c}_summary.txt 2>&1
  python3 tools/trace_timeline.py $out/trace_$c 5 >> $out/${c}_summary.txt 2>&1
  grep "^cfg" $out/${c}_log.txt >> $out/${c}_summary.txt
  rm -rf $out/trace_$c
  echo "$c traced"; head -12 $out/${c}_summary.txt
done
