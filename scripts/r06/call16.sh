#!/bin/bash
# round 6, GPU call 16: more fresh seeds on the final tree -- production path (every fuzz test, block loop forced, default hand-over) and the
# counting path with full-length re-trace of every shortened ray
O=gpurun_out/r06_16
mkdir -p $O
HZ_FUZZ_N=1300 HZ_FUZZ_SEED=68003 HZ_TEST_SCHEDULE="persist_grid=6" timeout 3000 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "random_configurations or adversarial_near or random_locations or random_terrain_shadow" --durations=4 2>&1 | tail -9 > $O/fuzz_68003_1300_all_persist_grid6.log
tail -3 $O/fuzz_68003_1300_all_persist_grid6.log
for seed in 69001 70001; do
  timeout 1200 python scripts/fuzz_near_adversarial.py --n 2600 --seed $seed --oracle-every 2 --out $O/tmp.jsonl 2> $O/fuzz_near_adversarial_seed${seed}_2600.summary.json
  tail -1 $O/fuzz_near_adversarial_seed${seed}_2600.summary.json | cut -c1-300
  grep PROBLEM $O/fuzz_near_adversarial_seed${seed}_2600.summary.json | head -3 | cut -c1-600
  rm -f $O/tmp.jsonl
done
