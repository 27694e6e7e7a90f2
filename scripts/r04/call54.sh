#!/bin/bash
# round 4, GPU call 54: the multi-rank lines again with the final kernels (two ranks on the one GPU over gloo; emulated 8-rank
# partition of config 5), smoke()
export TMPDIR=/tmp
O=gpurun_out/r04_54; mkdir -p $O
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 ); tail -1 $O/smoke.log
( HZ_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --no-cpu-baseline > $O/bench_c3_2ranks_one_gpu_gloo.json 2> $O/b2.err ); tail -1 $O/bench_c3_2ranks_one_gpu_gloo.json | cut -c1-300
( HZ_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --no-cpu-baseline --bcast verts --no-count --no-peaks > $O/bench_c3_2ranks_verts.json 2> $O/b2v.err ); tail -1 $O/bench_c3_2ranks_verts.json | cut -c1-200
( MASTER_ADDR=127.0.0.1 MASTER_PORT=29661 timeout 1500 python bench.py --workload c5 --emulate-ranks 8 > $O/c5_emulate8.json 2> $O/e8.err ); tail -1 $O/c5_emulate8.json | cut -c1-300
