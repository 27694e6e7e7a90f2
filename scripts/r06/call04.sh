#!/bin/bash
# round 6, GPU call 4: follow-up launch on the fast stack (groups that overflow are repeated), probes / env switches removed (same device code),
# bad-map bit index: parity + shadow + prep tests, then the same-box A/B against commit b7aacc5 (follow-up launch on the level stack)
O=gpurun_out/r06_04
mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_c4_shadow.py tests/test_gpu_prep.py tests/test_gpu_near_guard.py -x -q -m gpu 2>&1 | tail -8 > $O/tests.log
tail -3 $O/tests.log
for rep in 1 2 3; do
  (cd ab_old && timeout 300 python scripts/quick_perf.py --win 3569 --reps 2 2>&1 | grep -E "^rep|left" | sed "s/^/b7aacc5 rep$rep /") >> $O/ab.log
  for L in 0x20 0x24 0x28; do
    timeout 300 python scripts/quick_perf.py --win 3569 --reps 2 --left $L 2>&1 | grep -E "^rep|left" | sed "s/^/new left=$L rep$rep /" >> $O/ab.log
  done
done
grep -E "rep 1 wall|left " $O/ab.log | awk '{ if ($0 ~ /wall/) printf "%s %s %s | kernel %s ", $1,$2,$3,$9; else print $0 }' | sed 's/stack redo blocks 0  fallbacks 0//'
