#!/bin/bash
# round 4, GPU call 17: certificate pre-pass with every S-th azimuth plane evaluated (HZ_NEAR_STRIDE): cost, looseness, re-trace
export TMPDIR=/tmp
O=gpurun_out/r04_17; mkdir -p $O
for S in 1 2 3 4 1 2 3 4; do
  ( HZ_NEAR_STRIDE=$S timeout 300 python scripts/quick_perf.py --win 1024 --reps 4 --count --verify-near > $O/q.tmp 2>&1 ); echo "stride $S: $(grep -A1 'rep 2\|rep 3' $O/q.tmp | grep -v '^--' | awk '{printf "%s %s %s %s | ", $5,$6,$7,$8}')" >> $O/stride.log
done
cat $O/stride.log
for S in 1 2 3; do
  ( HZ_NEAR_STRIDE=$S timeout 300 python bench.py --steps 3 --no-extras --no-e2e --no-cpu-baseline --no-count --no-peaks > $O/b.tmp 2>/dev/null ); echo "stride $S whole tile: $(tail -1 $O/b.tmp | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_per_launch"], d["config"]["near_prepass_ms_per_step"], d["value"])')" >> $O/stride_tile.log
done
cat $O/stride_tile.log
( HZ_NEAR_STRIDE=2 timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_near_guard.py tests/test_gpu_parity.py -x -q -k "not stray" > $O/tests_s2.log 2>&1 ); tail -3 $O/tests_s2.log
( HZ_NEAR_STRIDE=3 timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_near_guard.py -x -q -k "not stray" > $O/tests_s3.log 2>&1 ); tail -3 $O/tests_s3.log
