#!/bin/bash
# round 5, GPU call 21: hit-cache ancestor level 3 / 4 / 5 (product) again, interleaved; config 4 with the pinned hz_crmath.h coefficients as the
# product (parity tests of the shadow path + timing)
export TMPDIR=/tmp
O=gpurun_out/r05_21; mkdir -p $O
( time timeout 600 python -c "import torch; print(torch.__version__)" ) > $O/torch_import.log 2>&1
for rep in 1 2 3; do
for lib in anc3 anc4 product; do
  if [ $lib = product ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_$lib.so; fi
  ( timeout 300 python scripts/quick_perf.py --win 3569 --reps 2 > $O/qp_${lib}_$rep.log 2>&1 ); echo qp $lib $rep $(grep "^rep 1" $O/qp_${lib}_$rep.log | cut -c1-120)
done
done
unset HORAYZON_HIP_LIB
( timeout 900 python -m pytest tests/test_gpu_c4_shadow.py tests/test_gpu_parity.py -x -q -k "shadow or refrac or terrain or c4" > $O/tests_shadow.log 2>&1 ); tail -3 $O/tests_shadow.log
for rf in 1 0 1 0; do
  ( timeout 300 python bench.py --workload c4 --refrac $rf > $O/c4_product_rf${rf}.json 2> $O/c4_product_rf${rf}.err ); echo c4 product refrac $rf $(python -c "import json; d=json.loads(open('$O/c4_product_rf${rf}.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)
done
for lib in anc4 product; do
  if [ $lib = product ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_$lib.so; fi
  ( timeout 300 python scripts/quick_perf.py --win 3569 --reps 2 --alg binary_search > $O/qp_bin_${lib}.log 2>&1 ); echo qp binary $lib $(grep "^rep 1" $O/qp_bin_${lib}.log | cut -c1-120)
  ( timeout 300 python scripts/quick_perf.py --win 1024 --reps 2 --alg discrete_sampling > $O/qp_dis_${lib}.log 2>&1 ); echo qp discrete $lib $(grep "^rep 1" $O/qp_dis_${lib}.log | cut -c1-120)
done
