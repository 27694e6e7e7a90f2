#!/bin/bash
# round 5, GPU call 1: box tests over [-tau, tfar + tau] -- the counter-example replay, the GPU suite, and a same-box A/B
# of the whole C3 tile against the round-4 library
export TMPDIR=/tmp
O=gpurun_out/r05_01; mkdir -p $O
( timeout 300 python scripts/replay_adv.py 48001 2536 > $O/replay_adv_48001_2536.log 2>&1 ); cat $O/replay_adv_48001_2536.log
( timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $O/tests_gpu.log 2>&1 ); tail -3 $O/tests_gpu.log
for rep in 1 2; do
  for lib in r4 product; do
    if [ $lib = product ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_$lib.so; fi
    ( timeout 300 python scripts/quick_perf.py --win 3569 --reps 3 > $O/perf_${lib}_$rep.log 2>&1 ); echo $lib $rep; grep "^rep" $O/perf_${lib}_$rep.log | cut -c1-120
  done
done
unset HORAYZON_HIP_LIB
