#!/bin/bash
# round 4, GPU call 47: final code: rocprofv3 passes (r04t), lines of record, whole GPU suite, a wide random sweep
export TMPDIR=/tmp
O=gpurun_out/r04_47; mkdir -p $O
bash scripts/profile_bench.sh r04t > $O/profile.log 2>&1; tail -1 $O/profile.log | cut -c1-200
rm -f gpurun_out/r04_near_verify.jsonl
( timeout 2700 python -m pytest tests -m gpu -q > $O/tests_gpu.log 2>&1 ); tail -4 $O/tests_gpu.log
cp gpurun_out/r04_near_verify.jsonl $O/ 2>/dev/null
( HZ_FUZZ_N=300 HZ_FUZZ_SEED=45001 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -x -k "not stray" > $O/fuzz_45001.log 2>&1 ); tail -1 $O/fuzz_45001.log
