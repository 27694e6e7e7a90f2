#!/bin/bash
# round 4, GPU call 1: new box decode (v_perm + v_fma_mix) -- parity suite, same-box A/B against the round-3 library, issue rates
export TMPDIR=/tmp
O=gpurun_out/r04_1; mkdir -p $O
R3=horayzon_amd/libhorayzon_hip_r3.so
( timeout 300 python scripts/inst_rates.py > $O/inst_rates.json 2> $O/inst_rates.err )
for v in r3 new r3 new; do
  if [ $v = r3 ]; then export HORAYZON_HIP_LIB=$PWD/$R3; else unset HORAYZON_HIP_LIB; fi
  ( timeout 600 python scripts/quick_perf.py --win 1024 --reps 3 --count >> $O/quick_$v.log 2>&1 )
done
unset HORAYZON_HIP_LIB
( timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1 ); tail -5 $O/tests_gpu.log
grep -h "rep 1\|rep 2\|SIMT" $O/quick_*.log
