#!/bin/bash
# round 6, GPU call 1: parity of the multi-level leftover hand-over, then the same-box A/B against the round-5 tree (ab_old/)
set -o pipefail
mkdir -p gpurun_out/r06_01
O=gpurun_out/r06_01
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -15 > $O/tests_parity.log
echo "parity rc=$?" >> $O/tests_parity.log
tail -3 $O/tests_parity.log
for rep in 1 2; do
  (cd ab_old && timeout 300 python scripts/quick_perf.py --win 3569 --reps 2 2>&1 | grep -E "^rep|left" | sed "s/^/r5 rep$rep /") >> $O/ab.log
  for L in 0 0x10 0x1010 0x10101010 0x2020 0x202020 0x181818 -1; do
    timeout 300 python scripts/quick_perf.py --win 3569 --reps 2 --left $L 2>&1 | grep -E "^rep|left" | sed "s/^/new left=$L rep$rep /" >> $O/ab.log
  done
done
cat $O/ab.log
