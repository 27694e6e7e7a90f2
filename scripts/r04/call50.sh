#!/bin/bash
# round 4, GPU call 50: rocprofv3 passes and lines of record with the class rates priced at their fastest sample (r04v)
export TMPDIR=/tmp
O=gpurun_out/r04_50; mkdir -p $O
bash scripts/profile_bench.sh r04v > $O/profile.log 2>&1; tail -1 $O/profile.log | cut -c1-200
( timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_line.err ); tail -1 $O/bench_line.json | cut -c1-300
( timeout 300 python bench.py --workload c4 > $O/bench_c4_shadow.json 2> $O/bench_c4.err ); tail -1 $O/bench_c4_shadow.json | cut -c1-300
( timeout 300 python bench.py --workload c4 --which sw_dir_cor --refrac 1 > $O/bench_c4_sw_dir_cor_refrac.json 2>> $O/bench_c4.err ); tail -1 $O/bench_c4_sw_dir_cor_refrac.json | cut -c1-200
