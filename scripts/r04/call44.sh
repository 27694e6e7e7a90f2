#!/bin/bash
# round 4, GPU call 44: shadow kernel (fast stack) held to 6 / 7 workgroups per CU
export TMPDIR=/tmp
O=gpurun_out/r04_45; mkdir -p $O
for round in 1 2; do
for v in sw7:19 sw8:17 sw8:15; do
  lib=${v%%:*}; cap=${v##*:}
  if [ $lib = new ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=$PWD/horayzon_amd/libhorayzon_hip_$lib.so; fi
  ( HZ_SHADOW_FAST_CAP=$cap timeout 200 python bench.py --workload c4 --no-peaks --no-count > $O/c4.tmp 2>$O/c4.err ); echo "$lib cap $cap: $(tail -1 $O/c4.tmp | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["ms_per_sun_position"])')" >> $O/ab_shadow_wg.log
done
done
cat $O/ab_shadow_wg.log
