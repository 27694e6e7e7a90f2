#!/bin/bash
# round 6, GPU call 21: one more pair of fresh seeds on the final tree (production path of every fuzz test; counting path with full re-trace)
O=gpurun_out/r06_21
mkdir -p $O
HZ_FUZZ_N=1000 HZ_FUZZ_SEED=71003 HZ_TEST_SCHEDULE="persist_grid=7" timeout 3000 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "random_configurations or adversarial_near or random_locations or random_terrain_shadow" 2>&1 | tail -3 > $O/fuzz_71003_1000_all_persist_grid7.log
tail -1 $O/fuzz_71003_1000_all_persist_grid7.log
timeout 1200 python scripts/fuzz_near_adversarial.py --n 2600 --seed 72001 --oracle-every 2 --out $O/tmp.jsonl 2> $O/fuzz_near_adversarial_seed72001_2600.summary.json
tail -1 $O/fuzz_near_adversarial_seed72001_2600.summary.json | cut -c1-300; rm -f $O/tmp.jsonl
