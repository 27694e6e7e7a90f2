#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r04_10; mkdir -p $O
( timeout 600 python scripts/r04/plain_vs_relief.py > $O/plain_vs_relief.jsonl 2> $O/pvr.err ); cat $O/plain_vs_relief.jsonl
for thr in 32 40 48; do for bias in 14 17 20 24 28; do
  r=$((thr + bias * 256))
  ( timeout 300 python scripts/quick_perf.py --win 1024 --reps 3 --regroup $r > $O/rg.tmp 2>&1 ); echo "regroup $thr bias $bias: $(grep 'rep 1\|rep 2' $O/rg.tmp | awk '{print $6}' | tr '\n' ' ')" >> $O/regroup_sweep.log
done; done
cat $O/regroup_sweep.log
