#!/bin/bash
# round 4, GPU call 5: 48 B nodes (3 loads per visit) against 64 B nodes on the same box; cost of the sampled certificate check
export TMPDIR=/tmp
O=gpurun_out/r04_5; mkdir -p $O
N64=$PWD/horayzon_amd/libhorayzon_hip_n64.so
for v in n64 n48 n64 n48; do
  if [ $v = n64 ]; then export HORAYZON_HIP_LIB=$N64; else unset HORAYZON_HIP_LIB; fi
  ( timeout 600 python scripts/quick_perf.py --win 1024 --reps 4 --count >> $O/quick_$v.log 2>&1 )
done
unset HORAYZON_HIP_LIB
for n in 0 1073741824 256 0 1073741824 256; do
  ( timeout 300 python scripts/quick_perf.py --win 1024 --reps 3 --verify-sample $n > $O/verify_$n.tmp 2>&1 ); echo "verify-sample $n" >> $O/verify_sample.log; grep "rep 1\|rep 2\|re-traced" $O/verify_$n.tmp >> $O/verify_sample.log
done
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_near_guard.py tests/test_gpu_prep.py -x -q -k "not stray" > $O/tests.log 2>&1 ); tail -4 $O/tests.log
grep -h "rep 2\|rep 3\|SIMT" $O/quick_n64.log $O/quick_n48.log; cat $O/verify_sample.log
