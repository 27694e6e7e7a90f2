#!/bin/bash
# round 4, GPU call 62: the balance test with its new thresholds
export TMPDIR=/tmp
O=gpurun_out/r04_62; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_bench_ranks.py -q -k "inhomogeneous" > $O/t.log 2>&1 ); tail -2 $O/t.log
( timeout 600 python -m pytest tests/test_gpu_bench_ranks.py -q -k "inhomogeneous" > $O/t2.log 2>&1 ); tail -2 $O/t2.log
