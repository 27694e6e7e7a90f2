#!/bin/bash
# round 6, GPU call 5: final kernels (follow-up launch on the fast stack, threshold 36, probes and env switches removed): full GPU suite,
# then the rocprofv3 passes of scripts/profile_bench.sh (kernel trace, FETCH / WRITE, SQ counters, wait / busy split; c3 and c4)
O=gpurun_out/r06_05
mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu --durations=5 2>&1 | tail -15 > $O/tests_gpu.log
tail -4 $O/tests_gpu.log
bash scripts/profile_bench.sh r06a > $O/profile_bench.log 2>&1
tail -2 $O/profile_bench.log | cut -c1-600
