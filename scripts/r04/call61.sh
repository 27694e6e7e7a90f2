#!/bin/bash
# round 4, GPU call 61: the whole GPU suite and the default bench line on the final tree (after the pre-pass change)
export TMPDIR=/tmp
O=gpurun_out/r04_61; mkdir -p $O
rm -f gpurun_out/r04_near_verify.jsonl
( timeout 2700 python -m pytest tests -m gpu -q > $O/tests_gpu.log 2>&1 ); tail -4 $O/tests_gpu.log
cp gpurun_out/r04_near_verify.jsonl $O/ 2>/dev/null
( timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_line.err ); tail -1 $O/bench_line.json | cut -c1-300
