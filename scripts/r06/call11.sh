#!/bin/bash
# round 6, GPU call 11: the certificate fix (frame-offset terms in the crossing interval of hz_near.hip): the two replayed configurations, the
# near-field tests, then adversarial sweeps (counting path with full re-trace + production path) on three seeds incl. the failing one, and its cost
O=gpurun_out/r06_11
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_near_guard.py tests/test_gpu_parity.py -x -q -m gpu -k "near or certificates or spike or c2_gaussian" 2>&1 | tail -5 > $O/tests_near.log
tail -3 $O/tests_near.log
timeout 900 python scripts/r06/replay_64003.py 2600 --all > $O/replay_64003_after_fix.log 2>&1
tail -3 $O/replay_64003_after_fix.log | cut -c1-600
for seed in 65001 66001; do
  timeout 1200 python scripts/fuzz_near_adversarial.py --n 2600 --seed $seed --oracle-every 2 --out $O/fuzz_near_adversarial_seed${seed}_2600.jsonl 2> $O/fuzz_near_adversarial_seed${seed}_2600.summary.json
  tail -1 $O/fuzz_near_adversarial_seed${seed}_2600.summary.json | cut -c1-400
  rm -f $O/fuzz_near_adversarial_seed${seed}_2600.jsonl
done
timeout 300 python scripts/quick_perf.py --win 3569 --reps 2 2>&1 | grep -E "^rep|near|left" > $O/perf_after_fix.log
cat $O/perf_after_fix.log
