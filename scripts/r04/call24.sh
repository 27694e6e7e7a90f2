#!/bin/bash
# round 4, GPU call 24: (sin, cos) pair tables and the near-certificate register cache against the previous commit
export TMPDIR=/tmp
O=gpurun_out/r04_24; mkdir -p $O
for v in prev new nc prev new nc; do
  if [ $v = new ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=$PWD/horayzon_amd/libhorayzon_hip_$v.so; fi
  ( timeout 300 python scripts/quick_perf.py --win 1024 --reps 4 > $O/q.tmp 2>&1 ); echo "$v $(grep 'rep 1\|rep 2\|rep 3' $O/q.tmp | awk '{print $6}' | tr '\n' ' ')" >> $O/ab_tables2.log
done
unset HORAYZON_HIP_LIB
cat $O/ab_tables2.log
