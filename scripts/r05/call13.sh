#!/bin/bash
# round 5, GPU call 13: the whole GPU suite on the current tree (threshold 36, k_topo plane branch in float32, monitor stores to a
# scratch row, POOL / probe code behind flags); shadow compaction threshold 40 / 28 / 24 / 20 with repeats; locations kernel
# threshold 40 / 36 / 32; whole tile against r4
export TMPDIR=/tmp
O=gpurun_out/r05_13; mkdir -p $O
( time timeout 600 python -c "import torch; print(torch.__version__)" ) > $O/torch_import.log 2>&1
( timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 > $O/tests_gpu.log 2>&1 ); tail -12 $O/tests_gpu.log
for rep in 1 2; do
for lib in product sr28 sr24 sr20; do
  if [ $lib = product ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_$lib.so; fi
  for rf in 0 1; do
    ( timeout 300 python bench.py --workload c4 --refrac $rf > $O/c4_${lib}_refrac${rf}_$rep.json 2> $O/c4_${lib}_refrac${rf}_$rep.err ); echo c4 $lib refrac $rf rep $rep $(python -c "import json; d=json.loads(open('$O/c4_${lib}_refrac${rf}_$rep.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)
  done
done
done
for lib in product lr36 lr32; do
  if [ $lib = product ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_$lib.so; fi
  ( timeout 300 python scripts/quick_locations.py --reps 3 > $O/loc_$lib.log 2>&1 ); echo locations $lib $(grep "^rep" $O/loc_$lib.log | sed 's/.*kernel \([0-9.]*\)s.*/\1/' | tr '\n' ' ')
done
for lib in r4 product; do
  if [ $lib = product ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_$lib.so; fi
  ( timeout 300 python scripts/quick_perf.py --win 3569 --reps 3 > $O/perf_${lib}.log 2>&1 ); echo whole $lib $(grep "^rep" $O/perf_${lib}.log | sed 's/.*kernel \([0-9.]*\)s.*/\1/' | tr '\n' ' ')
done
