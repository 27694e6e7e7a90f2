#!/bin/bash
# round 4, GPU call 25 (32 B nodes): rocprofv3 passes over bench.py, whole GPU suite, a wide random sweep, the lines of record
export TMPDIR=/tmp
O=gpurun_out/r04_25; mkdir -p $O
rm -f gpurun_out/r04_near_verify.jsonl
bash scripts/profile_bench.sh r04q > $O/profile.log 2>&1; tail -2 $O/profile.log | cut -c1-300
( timeout 2700 python -m pytest tests -m gpu -q > $O/tests_gpu.log 2>&1 ); tail -4 $O/tests_gpu.log
( HZ_FUZZ_N=300 HZ_FUZZ_SEED=44001 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -x -k "not stray" > $O/fuzz_44001.log 2>&1 ); tail -1 $O/fuzz_44001.log
cp gpurun_out/r04_near_verify.jsonl $O/ 2>/dev/null
