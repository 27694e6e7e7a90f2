#!/bin/bash
# round 4, GPU call 53: node loads with the array base in scalar registers (one address instruction instead of three)
export TMPDIR=/tmp
O=gpurun_out/r04_53; mkdir -p $O
for round in 1 2 3; do
for v in new sad; do
  if [ $v = new ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=$PWD/horayzon_amd/libhorayzon_hip_$v.so; fi
  ( timeout 120 python scripts/quick_perf.py --win 1024 --reps 3 > $O/q.tmp 2>&1 ); echo "$v $(grep 'rep 1\|rep 2' $O/q.tmp | awk '{print $6}' | tr '\n' ' ')" >> $O/ab.log
done
done
cat $O/ab.log
