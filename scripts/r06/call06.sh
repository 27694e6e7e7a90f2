#!/bin/bash
# round 6, GPU call 6: tuning of the follow-up launch (hz_opts.left_tune: compaction threshold, width of the azimuths-left classes of the sort key)
O=gpurun_out/r06_06
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_prep.py -x -q -m gpu -k "leftover or traversal_stack or topo or svf or sky" 2>&1 | tail -5 > $O/tests.log
tail -3 $O/tests.log
for rep in 1 2; do
  for T in 0x0000 0x0010 0x0018 0x0030 0x0300 0x0400 0x0500 0x0600 0x0418 0x0410; do
    timeout 300 python scripts/quick_perf.py --win 3569 --reps 2 --tune $T 2>&1 | grep -E "^rep|left" | sed "s/^/tune=$T rep$rep /" >> $O/ab.log
  done
done
grep -E "rep 1 wall|left " $O/ab.log | awk '{ if ($0 ~ /wall/) printf "%s %s | kernel %s ", $1,$2,$8; else print $0 }' | sed 's/stack redo blocks 0  fallbacks 0//'
