#!/bin/bash
# round 4, GPU call 3: certificate work -- new checks / rigorous crossing bounds in k_near_cert, sampled verification in the
# production kernel, full-size re-trace tests, adversarial sweep, pre-pass timing
export TMPDIR=/tmp
O=gpurun_out/r04_3; mkdir -p $O
rm -f gpurun_out/r04_near_verify.jsonl
( timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_near_guard.py -x -q -k "not stray" > $O/tests_fuzz.log 2>&1 ); tail -5 $O/tests_fuzz.log
( timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_bench_ranks.py -x -q -s > $O/tests_full.log 2>&1 ); tail -5 $O/tests_full.log
( timeout 1200 python scripts/fuzz_near_adversarial.py --n 1500 --seed 41001 --out $O/fuzz_near_41001.jsonl 2> $O/fuzz_near_41001.err ); tail -2 $O/fuzz_near_41001.err
( timeout 300 python scripts/quick_perf.py --win 1024 --reps 3 > $O/quick.log 2>&1 ); grep "rep\|near" $O/quick.log
cp gpurun_out/r04_near_verify.jsonl $O/ 2>/dev/null
