#!/bin/bash
for t in "5=0" "5=10,6=1" "5=10,6=2" "5=10,6=3" "5=5,6=1" "5=0"; do
  A8_TUNE=$t python bench.py --steps 6 --warmup 2 --no-cpu-baseline --gemm-table 2>/tmp/gs.log | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$t', j['ms_per_step'], j['roofline']['achieved_serial'])"
  grep "M=6248 N=3072 K=768\|M=199992 N=512 K=1536\|M=6248 N=768 K=3072" /tmp/gs.log | cut -c1-100
done
