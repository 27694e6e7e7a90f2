#!/bin/bash
# round 4, GPU call 7: whole GPU suite (monitor = sampled counting launch), nodelet + fast stack probe, bench with the c5 extra
export TMPDIR=/tmp
O=gpurun_out/r04_7; mkdir -p $O
rm -f gpurun_out/r04_near_verify.jsonl
for top in -1 85 -1 85 341; do
  ( timeout 300 python scripts/quick_perf.py --win 1024 --reps 3 --top $top > $O/top_$top.tmp 2>&1 ); echo "top $top" >> $O/nodelet.log; grep "rep 1\|rep 2" $O/top_$top.tmp >> $O/nodelet.log
done
cat $O/nodelet.log
( timeout 300 python scripts/quick_perf.py --win 1024 --reps 3 --verify-sample 256 > $O/monitor.log 2>&1 ); grep "rep\|re-traced" $O/monitor.log | tail -4
( timeout 2700 python -m pytest tests -m gpu -q > $O/tests_gpu.log 2>&1 ); tail -8 $O/tests_gpu.log
( time timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.time
tail -1 $O/bench_default.json | cut -c1-400
cp gpurun_out/r04_near_verify.jsonl $O/ 2>/dev/null
