#!/bin/bash
# round 6, GPU call 2: leftover records SORTED (azimuths left, position) before each follow-up launch: parity, then threshold sweep
set -o pipefail
O=gpurun_out/r06_02
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -15 > $O/tests_parity.log
echo "parity rc=$?" >> $O/tests_parity.log
tail -3 $O/tests_parity.log
for rep in 1 2; do
  (cd ab_old && timeout 300 python scripts/quick_perf.py --win 3569 --reps 2 2>&1 | grep -E "^rep" | sed "s/^/r5 rep$rep /") >> $O/ab.log
  for L in 0x10 0x18 0x20 0x28 0x30 0x38 0x1010 0x1020 0x2020 0x1030 0x2030 0x102030 0x101010 -1; do
    timeout 300 python scripts/quick_perf.py --win 3569 --reps 2 --left $L 2>&1 | grep -E "^rep|left" | sed "s/^/new left=$L rep$rep /" >> $O/ab.log
  done
done
grep -E "rep 1 wall|left " $O/ab.log | awk '{ if ($0 ~ /wall/) printf "%s %s %s | kernel %s ", $1,$2,$3,$9; else print $0 }' | sed 's/stack redo blocks 0  fallbacks 0//'
