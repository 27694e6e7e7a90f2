#!/bin/bash
# round 6, GPU call 7 (final tree): full GPU suite, the plain bench line, then the rocprofv3 passes of scripts/profile_bench.sh
O=gpurun_out/r06_07
mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu --durations=5 2>&1 | tail -15 > $O/tests_gpu.log
tail -4 $O/tests_gpu.log
timeout 1500 python bench.py > $O/bench_line.json 2> $O/bench.err
tail -c 600 $O/bench_line.json
bash scripts/profile_bench.sh r06b > $O/profile_bench.log 2>&1
tail -1 $O/profile_bench.log | cut -c1-300
