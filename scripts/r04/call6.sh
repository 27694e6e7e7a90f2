#!/bin/bash
# round 4, GPU call 6: whole GPU suite on the 48 B node build + the default bench line
export TMPDIR=/tmp
O=gpurun_out/r04_6; mkdir -p $O
rm -f gpurun_out/r04_near_verify.jsonl
( timeout 2400 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1 ); tail -5 $O/tests_gpu.log
( time timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.time
tail -1 $O/bench_default.json | cut -c1-700
cp gpurun_out/r04_near_verify.jsonl $O/ 2>/dev/null
