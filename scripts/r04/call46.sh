#!/bin/bash
# round 4, GPU call 46: shadow kernel with the fast stack at 7 workgroups per CU: tests, bench lines, profile passes of config 4
export TMPDIR=/tmp
O=gpurun_out/r04_46; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_c4_shadow.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -k "not stray" > $O/tests.log 2>&1 ); tail -3 $O/tests.log
( HZ_SHADOW_FAST_CAP=6 timeout 900 python -m pytest tests/test_gpu_c4_shadow.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -k "shadow or sw_dir or random_conf" > $O/tests_cap6.log 2>&1 ); tail -3 $O/tests_cap6.log
( timeout 300 python bench.py --workload c4 > $O/bench_c4_shadow.json 2> $O/bench_c4.err ); tail -1 $O/bench_c4_shadow.json | cut -c1-300
( timeout 300 python bench.py --workload c4 --which sw_dir_cor --refrac 1 > $O/bench_c4_sw_dir_cor_refrac.json 2>> $O/bench_c4.err ); tail -1 $O/bench_c4_sw_dir_cor_refrac.json | cut -c1-200
