#!/bin/bash
# round 5, GPU call 12: experiment -DHZ_LEAF_POOL (the leaf step pools the wave's queued leaves over its lanes; rays in LDS).
# The pool costs 7 KB of LDS per workgroup, which the fast stack cannot give up at 5 workgroups per CU (call 11: 22 entries ->
# 132 blocks overflow and their redo launch costs 0.11 s), so the A/B runs at EQUAL occupancy: HZ_LDS_BUDGET = 38 KB for both
# libraries (4 workgroups per CU, 27 / 34 stack entries).  Parity suite with the variant library first.
export TMPDIR=/tmp
O=gpurun_out/r05_12; mkdir -p $O
export HZ_LDS_BUDGET=38912
export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_pool.so
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -k "not stray" > $O/tests_pool.log 2>&1 ); tail -5 $O/tests_pool.log
( timeout 300 python scripts/quick_perf.py --win 3569 --reps 3 --count > $O/perf_pool.log 2>&1 ); echo whole pool $(grep "^rep" $O/perf_pool.log | sed 's/.*kernel \([0-9.]*\)s.*/\1/' | tr '\n' ' ') $(grep SIMT $O/perf_pool.log) $(grep redo $O/perf_pool.log | tail -1)
for bias in 16 24 32 40 56; do
  for thr in 36; do
    rg=$((thr + bias * 256))
    ( timeout 300 python scripts/quick_perf.py --win 3569 --reps 3 --count --regroup $rg > $O/perf_pool_rg$rg.log 2>&1 ); echo pool thr $thr bias $bias $(grep "^rep" $O/perf_pool_rg$rg.log | sed 's/.*kernel \([0-9.]*\)s.*/\1/' | tr '\n' ' ') $(grep SIMT $O/perf_pool_rg$rg.log)
  done
done
unset HORAYZON_HIP_LIB
( timeout 300 python scripts/quick_perf.py --win 3569 --reps 3 --count > $O/perf_product_4wg.log 2>&1 ); echo whole product 4wg $(grep "^rep" $O/perf_product_4wg.log | sed 's/.*kernel \([0-9.]*\)s.*/\1/' | tr '\n' ' ') $(grep SIMT $O/perf_product_4wg.log) $(grep redo $O/perf_product_4wg.log | tail -1)
unset HZ_LDS_BUDGET
( timeout 300 python scripts/quick_perf.py --win 3569 --reps 3 > $O/perf_product.log 2>&1 ); echo whole product 5wg $(grep "^rep" $O/perf_product.log | sed 's/.*kernel \([0-9.]*\)s.*/\1/' | tr '\n' ' ')
