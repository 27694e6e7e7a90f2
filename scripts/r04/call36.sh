#!/bin/bash
# round 4, GPU call 36: wave-level loop exit (blocking leaf kept as its number) + quad test without early outs: parity, then speed
export TMPDIR=/tmp
O=gpurun_out/r04_36; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "not stray" > $O/tests.log 2>&1 ); tail -3 $O/tests.log
if grep -q failed $O/tests.log; then exit 0; fi
for round in 1 2; do
for v in c1 k6 new; do
  if [ $v = new ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=$PWD/horayzon_amd/libhorayzon_hip_$v.so; fi
  ( timeout 120 python scripts/quick_perf.py --win 1024 --reps 3 > $O/q.tmp 2>&1 ); echo "$v $(grep 'rep 1\|rep 2' $O/q.tmp | awk '{print $6}' | tr '\n' ' ')" >> $O/ab.log
done
done
unset HORAYZON_HIP_LIB
cat $O/ab.log
( timeout 1200 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_near_guard.py tests/test_gpu_c4_shadow.py tests/test_gpu_prep.py -q -k "not stray" > $O/tests2.log 2>&1 ); tail -3 $O/tests2.log
