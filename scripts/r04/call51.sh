#!/bin/bash
# round 4, GPU call 51: config-4 lines with the calibrated set-up constant (plain and under the rocprofv3 kernel trace)
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r04_51; mkdir -p $O
( timeout 300 python bench.py --workload c4 > $O/bench_c4_shadow.json 2> $O/bench_c4.err ); tail -1 $O/bench_c4_shadow.json | cut -c1-300
( timeout 300 python bench.py --workload c4 --which sw_dir_cor --refrac 1 > $O/bench_c4_sw_dir_cor_refrac.json 2>> $O/bench_c4.err ); tail -1 $O/bench_c4_sw_dir_cor_refrac.json | cut -c1-200
cd /tmp
rm -rf $R/gpurun_out/r04v_c4kt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04v_c4kt -- python $R/bench.py --workload c4 > $R/gpurun_out/r04v_c4kt_bench.json 2> $R/gpurun_out/r04v_c4kt.err
