#!/bin/bash
# round 4, GPU call 49: the lines of record on the final code + the new regression test
export TMPDIR=/tmp
O=gpurun_out/r04_49; mkdir -p $O
( timeout 300 python -m pytest tests/test_gpu_fuzz.py -q -k "survives" > $O/test_new.log 2>&1 ); tail -1 $O/test_new.log
( timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_line.err ); tail -1 $O/bench_line.json | cut -c1-300
( timeout 300 python bench.py --workload c4 > $O/bench_c4_shadow.json 2> $O/bench_c4.err ); tail -1 $O/bench_c4_shadow.json | cut -c1-300
( timeout 300 python bench.py --workload c4 --which sw_dir_cor --refrac 1 > $O/bench_c4_sw_dir_cor_refrac.json 2>> $O/bench_c4.err ); tail -1 $O/bench_c4_sw_dir_cor_refrac.json | cut -c1-200
