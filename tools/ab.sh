#!/bin/bash
# same-box A/B of bench.py under different env toggles: tools/ab.sh "A8_EARLY_NEG=0" "A8_EARLY_NEG=1" ...
for round in 1 2; do
  for cfg in "$@"; do
    ms=$(env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $round  $cfg  ->  $ms ms/step"
  done
done
