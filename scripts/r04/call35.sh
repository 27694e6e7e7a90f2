#!/bin/bash
# round 4, GPU call 34: the traversal loop left by the wave (no per-lane exit, no result register) against the commit before (c1)
export TMPDIR=/tmp
O=gpurun_out/r04_35; mkdir -p $O
for round in 1 2; do
for v in c1 k6 new un; do
  if [ $v = new ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=$PWD/horayzon_amd/libhorayzon_hip_$v.so; fi
  ( timeout 300 python scripts/quick_perf.py --win 1024 --reps 3 > $O/q.tmp 2>&1 ); echo "$v $(grep 'rep 1\|rep 2' $O/q.tmp | awk '{print $6}' | tr '\n' ' ')" >> $O/ab.log
done
done
unset HORAYZON_HIP_LIB
cat $O/ab.log
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_near_guard.py tests/test_gpu_c4_shadow.py -q -k "not stray" > $O/tests.log 2>&1 ); tail -3 $O/tests.log
