#!/bin/bash
for t in "5=0" "5=1" "5=400" "5=0" "5=1"; do
  A8_TUNE=$t python bench.py --steps 8 --warmup 3 --no-cpu-baseline --gemm-table 2>/tmp/gs.log | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$t', j['ms_per_step'], j['roofline']['achieved'], j['roofline']['achieved_serial'])"
  grep "gemm NT" /tmp/gs.log | head -14 | cut -c1-100
done
