#!/bin/bash
# round 4, GPU call 9: monitor on its own stream, new cost probe on the half-plain DEM, regroup / leaf-bias sweep on the new node step
export TMPDIR=/tmp
O=gpurun_out/r04_9; mkdir -p $O
rm -f gpurun_out/r04_near_verify.jsonl
for n in 0 256 0 256; do
  ( timeout 300 python scripts/quick_perf.py --win 1024 --reps 3 --verify-sample $n > $O/verify_$n.tmp 2>&1 ); echo "verify-sample $n" >> $O/monitor.log; grep "rep 1\|rep 2\|re-traced" $O/verify_$n.tmp >> $O/monitor.log
done
cat $O/monitor.log
for thr in 32 40 48; do for bias in 14 17 20 24 28; do
  r=$((thr + bias * 256))
  ( timeout 300 python scripts/quick_perf.py --win 1024 --reps 3 --regroup $r > $O/rg.tmp 2>&1 ); echo "regroup $thr bias $bias: $(grep 'rep 1\|rep 2' $O/rg.tmp | awk '{print $7}' | tr '\n' ' ')" >> $O/regroup_sweep.log
done; done
cat $O/regroup_sweep.log
( timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_bench_ranks.py -q -s -k "certificates or inhomogeneous" > $O/tests.log 2>&1 ); tail -5 $O/tests.log; grep "cost\b\|\"cost\"" $O/tests.log | head -3
cp gpurun_out/r04_near_verify.jsonl $O/ 2>/dev/null
