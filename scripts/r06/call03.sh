#!/bin/bash
# round 6, GPU call 3: default hand-over threshold 32 (sorted leftover records): full GPU suite, long fuzz sweeps under a forced block loop, bench line
O=gpurun_out/r06_03
mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu --durations=5 2>&1 | tail -15 > $O/tests_gpu.log
tail -4 $O/tests_gpu.log
HZ_FUZZ_N=1000 HZ_FUZZ_SEED=61003 HZ_TEST_SCHEDULE="persist_grid=5" timeout 1500 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "random_configurations or adversarial_near" --durations=3 2>&1 | tail -8 > $O/fuzz_61003_1000_persist_grid5.log
tail -3 $O/fuzz_61003_1000_persist_grid5.log
HZ_FUZZ_N=500 HZ_FUZZ_SEED=62003 HZ_TEST_SCHEDULE="persist_grid=3,left_min=0x0c1830" timeout 1500 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "random_configurations or adversarial_near" --durations=3 2>&1 | tail -8 > $O/fuzz_62003_500_grid3_levels3.log
tail -3 $O/fuzz_62003_500_grid3_levels3.log
timeout 1500 python bench.py > $O/bench_line.json 2> $O/bench.err
tail -c 1500 $O/bench_line.json
