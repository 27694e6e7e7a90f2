#!/bin/bash
# round 4, GPU call 19: stack pushes without vector compares / shift-adds against the previous library; parity tests
export TMPDIR=/tmp
O=gpurun_out/r04_19; mkdir -p $O
for v in prev new prev new prev new; do
  if [ $v = new ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=$PWD/horayzon_amd/libhorayzon_hip_$v.so; fi
  ( timeout 300 python scripts/quick_perf.py --win 1024 --reps 4 > $O/q.tmp 2>&1 ); echo "$v $(grep 'rep 1\|rep 2\|rep 3' $O/q.tmp | awk '{print $6}' | tr '\n' ' ')" >> $O/ab_stack_push.log
done
unset HORAYZON_HIP_LIB
cat $O/ab_stack_push.log
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_c4_shadow.py -x -q -k "not stray" > $O/tests.log 2>&1 ); tail -3 $O/tests.log
