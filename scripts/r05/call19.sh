#!/bin/bash
# round 5, GPU call 19: (a) hit-cache walks that find nothing continue with the root inside hz_trace (root pre-pushed below the cached
# subtree) against the old restart through the caller (-DHZ_V_CACHE_RESTART): whole-tile A/B, interleaved; (b) deg2rad / rad2deg with the
# constant divisions as x rc + fma corrections against IEEE divisions (-DHZ_V_IEEE_CONST_DIV): config 4 +- refraction; (c) parity tests
export TMPDIR=/tmp
O=gpurun_out/r05_19; mkdir -p $O
( time timeout 600 python -c "import torch; print(torch.__version__)" ) > $O/torch_import.log 2>&1
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_c4_shadow.py tests/test_gpu_fullsize.py -x -q > $O/tests_parity.log 2>&1 ); tail -3 $O/tests_parity.log
for rep in 1 2 3; do
for lib in cacheold product; do
  if [ $lib = product ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_$lib.so; fi
  ( timeout 300 python scripts/quick_perf.py --win 3569 --reps 2 > $O/qp_${lib}_$rep.log 2>&1 ); echo qp $lib $rep $(grep "^rep 1" $O/qp_${lib}_$rep.log | cut -c1-200)
done
done
unset HORAYZON_HIP_LIB
( timeout 300 python scripts/quick_perf.py --win 3569 --reps 2 --count > $O/qp_product_count.log 2>&1 ); tail -4 $O/qp_product_count.log | cut -c1-250
( HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_cacheold.so timeout 300 python scripts/quick_perf.py --win 3569 --reps 2 --count > $O/qp_cacheold_count.log 2>&1 ); tail -4 $O/qp_cacheold_count.log | cut -c1-250
for rep in 1 2; do
for lib in ieeediv product; do
for rf in 1 0; do
  if [ $lib = product ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_$lib.so; fi
  ( timeout 300 python bench.py --workload c4 --refrac $rf > $O/c4_${lib}_rf${rf}_$rep.json 2> $O/c4_${lib}_rf${rf}_$rep.err ); echo c4 $lib refrac $rf rep $rep $(python -c "import json; d=json.loads(open('$O/c4_${lib}_rf${rf}_$rep.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)
done
done
done
