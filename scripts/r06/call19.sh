#!/bin/bash
# round 6, GPU call 19: the full GPU suite on the final tree (8a318d6 + docs)
O=gpurun_out/r06_19
mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu --durations=5 2>&1 | tail -12 > $O/tests_gpu.log
tail -3 $O/tests_gpu.log
