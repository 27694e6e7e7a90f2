#!/bin/bash
# round 6, GPU call 12 (final tree, after the certificate fix): full GPU suite, the plain bench line, the rocprofv3 passes, and the sweep that
# failed in call 8 (seed 64003, two hand-over levels, block loop forced) once more
O=gpurun_out/r06_12
mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu --durations=5 2>&1 | tail -15 > $O/tests_gpu.log
tail -4 $O/tests_gpu.log
timeout 1500 python bench.py > $O/bench_line.json 2> $O/bench.err
tail -c 400 $O/bench_line.json
bash scripts/profile_bench.sh r06c > $O/profile_bench.log 2>&1
HZ_FUZZ_N=1300 HZ_FUZZ_SEED=64003 HZ_TEST_SCHEDULE="persist_grid=3,left_min=0x1020" timeout 2400 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "random_configurations or adversarial_near" --durations=3 2>&1 | tail -8 > $O/fuzz_64003_1300_2600_grid3_two_levels.log
tail -3 $O/fuzz_64003_1300_2600_grid3_two_levels.log
