#!/bin/bash
# round 5, GPU call 33: leftover cells with the hand-over at the top of the loop body (node step 76 VALU, no scratch) and the leftover
# launch as its own instantiation -- parity, then against the library of commit b0ac5de on the same box
export TMPDIR=/tmp
O=gpurun_out/r05_33; mkdir -p $O
( time timeout 900 python -c "import torch; print(torch.__version__)" ) > $O/torch_import.log 2>&1
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_near_guard.py -m gpu -x -q --durations=3 > $O/tests_parity.log 2>&1 ); tail -2 $O/tests_parity.log
for rep in 1 2; do
for cfg in "r5b 0" "new 16" "new 12" "new 20" "new 0"; do
  set -- $cfg
  if [ $1 = new ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_$1.so; fi
  ( HZ_LEFT_TRACE=1 HZ_LEFT_MIN=$2 timeout 300 python scripts/quick_perf.py --win 3569 --reps 2 > $O/whole_$1_$2_$rep.log 2>&1 ); echo whole $1 left_min $2 rep $rep $(grep "^rep" $O/whole_$1_$2_$rep.log | awk '{print $6}' | tr '\n' ' ') $(grep "leftover cells" $O/whole_$1_$2_$rep.log | tail -1)
done
done
for cfg in "r5b 0" "new 16"; do
  set -- $cfg
  if [ $1 = new ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_$1.so; fi
  ( HZ_LEFT_MIN=$2 timeout 300 python bench.py --rows-per-step 447 --steps 8 --warmup 3 --no-extras --no-e2e --no-cpu-baseline --no-count --no-peaks > $O/slab_$1.json 2> $O/slab_$1.err ); echo slab447 $1 left_min $2 $(python -c "import json; d=json.loads(open('$O/slab_$1.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms_per_launch'])" 2>&1 | tail -1)
done
