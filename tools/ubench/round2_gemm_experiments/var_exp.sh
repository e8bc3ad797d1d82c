#!/bin/bash
for t in "0=5" "0=6" "0=5" "0=6"; do
  A8_TUNE=$t python bench.py --steps 6 --warmup 2 --no-cpu-baseline --gemm-table 2>/tmp/gs.log | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$t', j['ms_per_step'], j['roofline']['achieved_serial'])"
  grep "M=199992\|M=200000\|M=99992\|M=100000\|K=199992\|K=99992\|M=6248 N=3072\|M=6248 N=768 K=3072\|TN M=-4" /tmp/gs.log | cut -c1-100
done
