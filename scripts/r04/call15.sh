#!/bin/bash
# round 4, GPU call 15: test durations; launch tail (whole tile against 8 slabs of 446 rows; per-XCD spans)
export TMPDIR=/tmp
O=gpurun_out/r04_15; mkdir -p $O
( timeout 2700 python -m pytest tests -m gpu -q --durations=40 > $O/tests_gpu_durations.log 2>&1 ); tail -50 $O/tests_gpu_durations.log
( timeout 600 python bench.py --rows-per-step 446 --steps 8 --warmup 1 --no-cpu-baseline --no-e2e --no-extras --no-count --no-peaks > $O/bench_slabs_446.json 2>/dev/null ); tail -1 $O/bench_slabs_446.json | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("446-row slabs:", d["ms_per_step"], d["roofline"]["kernel_ms_per_launch"], d["config"]["near_prepass_ms_per_step"], d["roofline"]["svf_kernel_ms_per_launch"], d["value"])'
( HZ_XCD_TRACE=1 timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-extras --no-peaks > /dev/null 2> $O/xcd_spans.log ); grep "xcd spans" $O/xcd_spans.log
