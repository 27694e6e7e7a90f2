#!/bin/bash
# round 5, GPU call 4: which parity test stopped making progress in call 3?  (per-test timeout, verbose)
export TMPDIR=/tmp
O=gpurun_out/r05_04; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -v --timeout 100 --durations=10 > $O/tests_parity_fuzz.log 2>&1 ); grep -E "PASSED|FAILED|Timeout|ERROR|passed|failed" $O/tests_parity_fuzz.log | tail -70 | cut -c1-160
