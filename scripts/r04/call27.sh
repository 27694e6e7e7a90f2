#!/bin/bash
# round 4, GPU call 27: what does the node step wait for?  Dead instructions of one kind per node step (hz_common.h: HZ_PROBE_PADS)
# against the product library, same box, 1024^2 probe, alternating.
export TMPDIR=/tmp
O=gpurun_out/r04_27; mkdir -p $O
for round in 1 2; do
for v in base slow8 fast8 salu8 lds2 vmem1 slow16; do
  if [ $v = base ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=$PWD/horayzon_amd/libhorayzon_hip_$v.so; fi
  ( timeout 300 python scripts/quick_perf.py --win 1024 --reps 3 > $O/q.tmp 2>&1 ); echo "$v $(grep 'rep 1\|rep 2' $O/q.tmp | awk '{print $6}' | tr '\n' ' ')" >> $O/sensitivity.log
done
done
unset HORAYZON_HIP_LIB
cat $O/sensitivity.log
