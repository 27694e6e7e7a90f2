#!/bin/bash
# round 4, GPU call 21: how much do the bounds of a 32 B node cost in node visits?  (variant n32b: same 48 B node format, looser bounds)
# and: 6 workgroups per CU (80 VGPRs, 22 stack entries)
export TMPDIR=/tmp
O=gpurun_out/r04_21; mkdir -p $O
for v in base n32b base n32b; do
  if [ $v = base ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=$PWD/horayzon_amd/libhorayzon_hip_$v.so; fi
  ( timeout 300 python scripts/quick_perf.py --win 1024 --reps 4 --count > $O/q.tmp 2>&1 ); echo "$v $(grep 'rep 1\|rep 2' $O/q.tmp | awk '{print $6}' | tr '\n' ' ') $(grep 'rep 3' $O/q.tmp | awk '{print $17,$18,$19,$20}') $(grep SIMT $O/q.tmp)" >> $O/node32_bounds.log
done
cat $O/node32_bounds.log
export HORAYZON_HIP_LIB=$PWD/horayzon_amd/libhorayzon_hip_wg6.so
for b in 31744 26624; do
  ( HZ_LDS_BUDGET=$b timeout 300 python scripts/quick_perf.py --win 1024 --reps 4 > $O/q.tmp 2>&1 ); echo "wg6 lds $b: $(grep 'rep 1\|rep 2\|rep 3' $O/q.tmp | awk '{print $6}' | tr '\n' ' ')" >> $O/wg6.log
done
unset HORAYZON_HIP_LIB
( HZ_LDS_BUDGET=26624 timeout 300 python scripts/quick_perf.py --win 1024 --reps 4 > $O/q.tmp 2>&1 ); echo "base (5 per CU by registers) lds 26624: $(grep 'rep 1\|rep 2\|rep 3' $O/q.tmp | awk '{print $6}' | tr '\n' ' ')" >> $O/wg6.log
cat $O/wg6.log
