#!/bin/bash
# round 4, GPU call 28: the fast stack rewritten for the fewest instructions (sentinel entry, top two entries in one
# ds_read2st64, unconditional push candidates, selects instead of exec-mask branches) against the previous commit; parity
export TMPDIR=/tmp
O=gpurun_out/r04_28; mkdir -p $O
for round in 1 2; do
for v in prev fs1 new; do
  if [ $v = new ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=$PWD/horayzon_amd/libhorayzon_hip_$v.so; fi
  ( timeout 300 python scripts/quick_perf.py --win 1024 --reps 4 --count > $O/q.tmp 2>&1 ); echo "$v $(grep 'rep 1\|rep 2' $O/q.tmp | awk '{print $6}' | tr '\n' ' ') count: $(grep 'rep 3' $O/q.tmp | awk '{print $6, $17,$18,$19,$20}') $(grep SIMT $O/q.tmp)" >> $O/ab_fast_stack.log
done
done
unset HORAYZON_HIP_LIB
cat $O/ab_fast_stack.log
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_near_guard.py tests/test_gpu_prep.py -x -q -k "not stray" > $O/tests.log 2>&1 ); tail -5 $O/tests.log
