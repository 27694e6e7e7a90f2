#!/bin/bash
# round 6, GPU call 20: after the validator refactor (Python only): smoke, parity, shadow and prep tests
O=gpurun_out/r06_20
mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > $O/smoke.log; cat $O/smoke.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_c4_shadow.py tests/test_gpu_prep.py tests/test_gpu_near_guard.py -x -q -m gpu 2>&1 | tail -3 > $O/tests.log; cat $O/tests.log
