#!/bin/bash
# round 5, GPU call 3: (a) the per-cell height-field guard (bad-quad bitmap) -- the guard tests incl. the C3 tile with steps, a
# NoData hole and folds; (b) does the POSITION of the traversal loop in the instruction stream matter?  The tau change cost
# 1.3 % with an identical hot loop (opcode for opcode) that merely moved by 40 bytes: sweep the shift in 8-byte steps.
export TMPDIR=/tmp
O=gpurun_out/r05_03; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_near_guard.py -x -q --durations=8 > $O/tests_near_guard.log 2>&1 ); tail -12 $O/tests_near_guard.log
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q > $O/tests_parity_fuzz.log 2>&1 ); tail -3 $O/tests_parity_fuzz.log
for rep in 1 2; do
  for lib in r4 product s2 s4 s6 s8 s10 s12 s14; do
    if [ $lib = product ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_$lib.so; fi
    ( timeout 300 python scripts/quick_perf.py --win 3569 --reps 3 > $O/perf_${lib}_$rep.log 2>&1 ); echo $lib $rep $(grep "^rep" $O/perf_${lib}_$rep.log | sed 's/.*kernel \([0-9.]*\)s.*/\1/' | tr '\n' ' ')
  done
done
unset HORAYZON_HIP_LIB
