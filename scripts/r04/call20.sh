#!/bin/bash
# round 4, GPU call 20: cache-line prefetch of the next link (variant pf) against the product library
export TMPDIR=/tmp
O=gpurun_out/r04_20; mkdir -p $O
for v in base pf base pf base pf; do
  if [ $v = base ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=$PWD/horayzon_amd/libhorayzon_hip_$v.so; fi
  ( timeout 300 python scripts/quick_perf.py --win 1024 --reps 4 > $O/q.tmp 2>&1 ); echo "$v $(grep 'rep 1\|rep 2\|rep 3' $O/q.tmp | awk '{print $6}' | tr '\n' ' ')" >> $O/ab_prefetch.log
done
cat $O/ab_prefetch.log
