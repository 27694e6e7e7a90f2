#!/bin/bash
# round 4, GPU call 2: bench.py --gpus N self-launch + strong-scaling shard tests, the default bench line with extras
export TMPDIR=/tmp
O=gpurun_out/r04_2; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_bench_ranks.py -x -q > $O/tests_bench_ranks.log 2>&1 ); tail -15 $O/tests_bench_ranks.log
( time timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.time; tail -c 3000 $O/bench_default.err
tail -1 $O/bench_default.json | cut -c1-1500
