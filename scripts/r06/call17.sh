#!/bin/bash
# round 6, GPU call 17: the targeted worst-case test of the certificate fix (frames skewed up to the refusal threshold beside 300 m / 1000 m spikes)
O=gpurun_out/r06_17
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_near_guard.py -x -q -m gpu --durations=3 2>&1 | tail -8 > $O/tests_near_guard.log
cat $O/tests_near_guard.log
