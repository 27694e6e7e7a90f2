#!/bin/bash
# round 6, GPU call 14: sort key of the follow-up launch = classes of the ESTIMATED work left (azimuths left x rays per azimuth so far, log scale) instead of azimuth classes
O=gpurun_out/r06_14
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "leftover or traversal_stack or c2_gaussian" 2>&1 | tail -3 > $O/tests.log
tail -2 $O/tests.log
for rep in 1 2; do
  for T in 0x000000 0x040000 0x080000 0x100000 0x200000 0x080010 0x100010; do
    timeout 300 python scripts/quick_perf.py --win 3569 --reps 2 --tune $T 2>&1 | grep -E "^rep|left" | sed "s/^/tune=$T rep$rep /" >> $O/ab.log
  done
done
grep -E "rep 1 wall|left " $O/ab.log | awk '{ if ($0 ~ /wall/) printf "%s %s | kernel %s ", $1,$2,$8; else print $0 }' | sed 's/stack redo blocks 0  fallbacks 0//' | grep kernel
