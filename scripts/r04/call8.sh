#!/bin/bash
# round 4, GPU call 8: cost-balance on the half-plain DEM (numbers), certificate pre-pass v3 (task map, adaptive CH) timing + reasons
export TMPDIR=/tmp
O=gpurun_out/r04_8; mkdir -p $O
( timeout 300 python scripts/quick_perf.py --win 1024 --reps 3 > $O/quick.log 2>&1 ); grep "rep\|near" $O/quick.log
( timeout 300 python scripts/r04/near_reasons_diag.py > /dev/null 2> $O/near_reasons.log ); tail -2 $O/near_reasons.log
( timeout 600 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_near_guard.py -x -q -k "not stray" > $O/tests_fuzz.log 2>&1 ); tail -3 $O/tests_fuzz.log
MASTER_ADDR=127.0.0.1 MASTER_PORT=29651 timeout 900 python bench.py --workload c5 --tile 2049 --azim 72 --plain-fraction 0.5 --emulate-ranks 2 --cost-samples 64 > $O/balance_2049.json 2> $O/balance.err; tail -1 $O/balance_2049.json | cut -c1-1800
