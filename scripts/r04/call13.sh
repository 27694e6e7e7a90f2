#!/bin/bash
# round 4, GPU call 13: k_topo at 4 workgroups per CU; adversarial sweeps on the final code (2 x 2600 configurations);
# plain `bench.py --gpus 2` at full size (two ranks on the one GPU, gloo); emulated 8-rank partitions of c3 and c5
export TMPDIR=/tmp
O=gpurun_out/r04_13; mkdir -p $O
for v in base topo4 base topo4; do
  if [ $v = base ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=$PWD/horayzon_amd/libhorayzon_hip_$v.so; fi
  ( timeout 300 python bench.py --steps 3 --no-extras --no-e2e --no-cpu-baseline --no-count --no-peaks > $O/t.tmp 2>/dev/null ); echo "$v svf_ms $(tail -1 $O/t.tmp | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["roofline"]["svf_kernel_ms_per_launch"], d["ms_per_step"])')" >> $O/topo_wg.log
done
unset HORAYZON_HIP_LIB
cat $O/topo_wg.log
( timeout 1500 python scripts/fuzz_near_adversarial.py --n 2600 --seed 42001 --out $O/fuzz_near_42001.jsonl 2> $O/fuzz_near_42001.err ); tail -1 $O/fuzz_near_42001.err
( timeout 1500 python scripts/fuzz_near_adversarial.py --n 2600 --seed 42002 --out $O/fuzz_near_42002.jsonl 2> $O/fuzz_near_42002.err ); tail -1 $O/fuzz_near_42002.err
( HZ_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --no-cpu-baseline > $O/bench_c3_2ranks_one_gpu_gloo.json 2> $O/b2.err ); tail -1 $O/bench_c3_2ranks_one_gpu_gloo.json | cut -c1-300
( HZ_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --no-cpu-baseline --bcast verts --no-count --no-peaks > $O/bench_c3_2ranks_verts.json 2> $O/b2v.err ); tail -1 $O/bench_c3_2ranks_verts.json | cut -c1-200
( MASTER_ADDR=127.0.0.1 MASTER_PORT=29661 timeout 1500 python bench.py --workload c5 --emulate-ranks 8 > $O/c5_emulate8.json 2> $O/e8.err ); tail -1 $O/c5_emulate8.json | cut -c1-300
