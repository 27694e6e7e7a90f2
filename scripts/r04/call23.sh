#!/bin/bash
# round 4, GPU call 23: packed tables (one 16 B load per elevation index, one 8 B load per azimuth) against the previous commit
export TMPDIR=/tmp
O=gpurun_out/r04_23; mkdir -p $O
for v in prev new prev new prev new; do
  if [ $v = new ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=$PWD/horayzon_amd/libhorayzon_hip_$v.so; fi
  ( timeout 300 python scripts/quick_perf.py --win 1024 --reps 4 > $O/q.tmp 2>&1 ); echo "$v $(grep 'rep 1\|rep 2\|rep 3' $O/q.tmp | awk '{print $6}' | tr '\n' ' ')" >> $O/ab_tables.log
done
unset HORAYZON_HIP_LIB
cat $O/ab_tables.log
for alg in binary_search discrete_sampling; do for v in prev new; do
  if [ $v = new ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=$PWD/horayzon_amd/libhorayzon_hip_$v.so; fi
  ( timeout 300 python scripts/quick_perf.py --win 512 --reps 3 --alg $alg > $O/q.tmp 2>&1 ); echo "$alg $v $(grep 'rep 1\|rep 2' $O/q.tmp | awk '{print $6}' | tr '\n' ' ')" >> $O/ab_tables.log
done; done
unset HORAYZON_HIP_LIB
tail -4 $O/ab_tables.log
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -k "not stray" > $O/tests.log 2>&1 ); tail -3 $O/tests.log
