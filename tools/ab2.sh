#!/bin/bash
# same-box A/B of bench.py under different developer knobs: tools/ab2.sh OUTDIR "ENV1" "ENV2" ...   (each ENV a string of VAR=val pairs)
out=$1; shift
mkdir -p $out
for rnd in 1 2; do
  i=0
  for e in "$@"; do
    i=$((i+1))
    env $e python bench.py --steps ${AB_STEPS:-10} --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=o['roofline']
print('round $rnd cfg $i [$e]: %.3f ms/step  gemm_serial %.1f TF (%.4f)  enc %.3f ms (%.3f)  conv %.3f ms (hbm %.3f mfma %.3f) conv0 fwd %.3f bwd %.3f' % (o['ms_per_step'], r['achieved_serial'], r['frac_serial'], r['encoder']['ms'], r['encoder']['frac_mfma'], r['conv_extractor']['ms'], r['conv_extractor']['frac_hbm'], r['conv_extractor']['frac_mfma'], r['conv_extractor']['layer0']['ms_fwd'], r['conv_extractor']['layer0']['ms_bwd']))" | tee -a $out/ab.log
  done
done
