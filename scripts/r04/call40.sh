#!/bin/bash
# round 4, GPU call 40: (index, value) pairs in the middle-index table against HEAD (c2); parity
export TMPDIR=/tmp
O=gpurun_out/r04_40; mkdir -p $O
for round in 1 2 3; do
for v in c2 new; do
  if [ $v = new ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=$PWD/horayzon_amd/libhorayzon_hip_$v.so; fi
  ( timeout 120 python scripts/quick_perf.py --win 1024 --reps 3 > $O/q.tmp 2>&1 ); echo "$v $(grep 'rep 1\|rep 2' $O/q.tmp | awk '{print $6}' | tr '\n' ' ')" >> $O/ab.log
done
done
unset HORAYZON_HIP_LIB
cat $O/ab.log
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x -k "not stray" > $O/tests.log 2>&1 ); tail -3 $O/tests.log
