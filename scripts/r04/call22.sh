#!/bin/bash
# round 4, GPU call 22: 32 B nodes (2 loads per visit, shared x / y ranges per half, 8-bit z, no register spills) against the
# 48 B node library of the previous commit; parity tests
export TMPDIR=/tmp
O=gpurun_out/r04_22; mkdir -p $O
for v in prev new prev new; do
  if [ $v = new ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=$PWD/horayzon_amd/libhorayzon_hip_$v.so; fi
  ( timeout 300 python scripts/quick_perf.py --win 1024 --reps 4 --count > $O/q.tmp 2>&1 ); echo "$v $(grep 'rep 1\|rep 2' $O/q.tmp | awk '{print $6}' | tr '\n' ' ') count: $(grep 'rep 3' $O/q.tmp | awk '{print $6, $17,$18,$19,$20}') $(grep SIMT $O/q.tmp)" >> $O/ab_node32.log
done
unset HORAYZON_HIP_LIB
cat $O/ab_node32.log
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_near_guard.py tests/test_gpu_c4_shadow.py tests/test_gpu_prep.py -x -q -k "not stray" > $O/tests.log 2>&1 ); tail -3 $O/tests.log
for v in prev new; do
  if [ $v = new ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=$PWD/horayzon_amd/libhorayzon_hip_$v.so; fi
  ( timeout 300 python bench.py --steps 3 --no-extras --no-e2e --no-cpu-baseline --no-count --no-peaks > $O/b.tmp 2>/dev/null ); echo "$v whole tile: $(tail -1 $O/b.tmp | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["kernel_ms_per_launch"], d["value"])')" >> $O/ab_node32_tile.log
  ( timeout 300 python bench.py --workload c4 --no-count --no-peaks > $O/c4.tmp 2>/dev/null ); echo "$v c4 ms per sun: $(tail -1 $O/c4.tmp | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["ms_per_sun_position"])')" >> $O/ab_node32_tile.log
done
cat $O/ab_node32_tile.log
