#!/bin/bash
# round 5, GPU call 9: the ray-compaction threshold (leave the traversal loop when fewer than `thr` lanes are traversing) and the
# leaf-vote bias re-swept on the whole C3 tile for the product and the -DHZ_TRI_FMA library (call 8: thr 32 beat the default 40
# by 1.4 % with the FMA leaf step); leaf queue of 1 / 3 entries; the k_topo change (plane-limited branch in float32)
export TMPDIR=/tmp
O=gpurun_out/r05_09; mkdir -p $O
( time timeout 600 python -c "import torch; print(torch.__version__)" ) > $O/torch_import.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_prep.py -x -q > $O/test_prep.log 2>&1 ); tail -3 $O/test_prep.log
for lib in product fma; do
  if [ $lib = product ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_$lib.so; fi
  for bias in 24 28; do
    for thr in 16 24 28 32 36 40; do
      rg=$((thr + bias * 256))
      ( timeout 300 python scripts/quick_perf.py --win 3569 --reps 2 --regroup $rg > $O/perf_${lib}_rg$rg.log 2>&1 ); echo $lib thr $thr bias $bias $(grep "^rep" $O/perf_${lib}_rg$rg.log | sed 's/.*kernel \([0-9.]*\)s.*/\1/' | tr '\n' ' ')
    done
  done
done
for lib in q1 q3; do
  export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_$lib.so
  ( timeout 300 python scripts/quick_perf.py --win 3569 --reps 2 > $O/perf_${lib}.log 2>&1 ); echo $lib $(grep "^rep" $O/perf_${lib}.log | sed 's/.*kernel \([0-9.]*\)s.*/\1/' | tr '\n' ' ')
done
unset HORAYZON_HIP_LIB
( timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-count --no-peaks --no-e2e --no-extras > $O/bench_short.json 2> $O/bench_short.err )
python -c "import json; d=json.loads(open('$O/bench_short.json').read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['roofline'].get('kernel_ms_per_launch'), d['roofline'].get('svf_kernel_ms_per_launch'), d['config'].get('near_prepass_ms_per_step'))"
