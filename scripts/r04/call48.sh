#!/bin/bash
# round 4, GPU call 48: overflow decided on node-step pointers only (a lane may enter with leaf links above the free entries):
# the wide sweep that found it, the whole suite, the rocprofv3 passes on the final code (r04u)
export TMPDIR=/tmp
O=gpurun_out/r04_48; mkdir -p $O
( HZ_FUZZ_N=300 HZ_FUZZ_SEED=45001 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -x -k "not stray" > $O/fuzz_45001.log 2>&1 ); tail -1 $O/fuzz_45001.log
bash scripts/profile_bench.sh r04u > $O/profile.log 2>&1; tail -1 $O/profile.log | cut -c1-200
rm -f gpurun_out/r04_near_verify.jsonl
( timeout 2700 python -m pytest tests -m gpu -q > $O/tests_gpu.log 2>&1 ); tail -4 $O/tests_gpu.log
cp gpurun_out/r04_near_verify.jsonl $O/ 2>/dev/null
