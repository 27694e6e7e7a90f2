#!/bin/bash
# round 4, GPU call 14: the lines of record (default bench, c4) with the calibrated model; smoke; whole GPU suite
export TMPDIR=/tmp
O=gpurun_out/r04_14; mkdir -p $O
( time timeout 1500 python bench.py > $O/bench_line.json 2> $O/bench.err ) 2> $O/bench.time; tail -3 $O/bench.time; tail -1 $O/bench_line.json | cut -c1-300
( timeout 600 python bench.py --workload c4 > $O/bench_c4_shadow.json 2> $O/c4.err ); tail -1 $O/bench_c4_shadow.json | cut -c1-300
( timeout 600 python bench.py --workload c4 --refrac 1 --which sw_dir_cor > $O/bench_c4_sw_dir_cor_refrac.json 2> $O/c4b.err )
( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 ); tail -1 $O/smoke.log
rm -f gpurun_out/r04_near_verify.jsonl
( timeout 2700 python -m pytest tests -m gpu -q > $O/tests_gpu.log 2>&1 ); tail -4 $O/tests_gpu.log
