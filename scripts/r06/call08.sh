#!/bin/bash
# round 6, GPU call 8 (final tree, commit 85536e0): long sweeps against the oracle -- 2 x (1300 random + 2600 adversarial) configurations with the
# persistent block loop forced onto their small grids (hand-over on: default schedule, and two hand-over levels)
O=gpurun_out/r06_08
mkdir -p $O
HZ_FUZZ_N=1300 HZ_FUZZ_SEED=63003 HZ_TEST_SCHEDULE="persist_grid=5" timeout 2400 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "random_configurations or adversarial_near" --durations=3 2>&1 | tail -8 > $O/fuzz_63003_1300_2600_persist_grid5.log
tail -3 $O/fuzz_63003_1300_2600_persist_grid5.log
HZ_FUZZ_N=1300 HZ_FUZZ_SEED=64003 HZ_TEST_SCHEDULE="persist_grid=3,left_min=0x1020" timeout 2400 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "random_configurations or adversarial_near" --durations=3 2>&1 | tail -8 > $O/fuzz_64003_1300_2600_grid3_two_levels.log
tail -3 $O/fuzz_64003_1300_2600_grid3_two_levels.log
