#!/bin/bash
# round 6, GPU call 13 (final tree ee3639a): fresh-seed sweeps of every fuzz test (gridded, adversarial, locations, shadow), default schedule with the block loop forced
O=gpurun_out/r06_13
mkdir -p $O
HZ_FUZZ_N=900 HZ_FUZZ_SEED=67003 HZ_TEST_SCHEDULE="persist_grid=4" timeout 3000 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "random_configurations or adversarial_near or random_locations or random_terrain_shadow" --durations=4 2>&1 | tail -9 > $O/fuzz_67003_900_all_persist_grid4.log
tail -3 $O/fuzz_67003_900_all_persist_grid4.log
