#!/bin/bash
# round 4, GPU call 38: vote re-swept after the loop got cheaper; parity of the rotated loop; whole suite; profile passes (r04r)
export TMPDIR=/tmp
O=gpurun_out/r04_38; mkdir -p $O
for rg in 32 40; do for bias in 20 24 28 32 40; do
  ( timeout 120 python scripts/quick_perf.py --win 1024 --reps 3 --regroup $((bias*256+rg)) > $O/q.tmp 2>&1 ); echo "regroup $rg bias $bias: $(grep 'rep 1\|rep 2' $O/q.tmp | awk '{print $6}' | tr '\n' ' ')" >> $O/regroup_sweep.log
done; done
cat $O/regroup_sweep.log
rm -f gpurun_out/r04_near_verify.jsonl
( timeout 2700 python -m pytest tests -m gpu -q > $O/tests_gpu.log 2>&1 ); tail -4 $O/tests_gpu.log
cp gpurun_out/r04_near_verify.jsonl $O/ 2>/dev/null
