#!/bin/bash
# round 4, GPU call 56: the whole GPU suite on the final tree
export TMPDIR=/tmp
O=gpurun_out/r04_56; mkdir -p $O
rm -f gpurun_out/r04_near_verify.jsonl
( timeout 2700 python -m pytest tests -m gpu -q --durations=8 > $O/tests_gpu.log 2>&1 ); tail -14 $O/tests_gpu.log
cp gpurun_out/r04_near_verify.jsonl $O/ 2>/dev/null
