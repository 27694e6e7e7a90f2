#!/bin/bash
# round 4, GPU call 43: shadow kernel with the fast stack (in-kernel retry of overflowed rays) against the level stack
export TMPDIR=/tmp
O=gpurun_out/r04_43; mkdir -p $O
for round in 1 2; do
for cap in 0 27 21 17; do
  ( HZ_SHADOW_FAST_CAP=$cap timeout 200 python bench.py --workload c4 --no-peaks > $O/c4.tmp 2>$O/c4.err ); echo "fast_cap $cap: $(tail -1 $O/c4.tmp | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["ms_per_sun_position"], d["roofline"].get("nodes_per_ray"), d["roofline"].get("tris_per_ray"))')" >> $O/ab_shadow_fast.log
done
done
cat $O/ab_shadow_fast.log
( HZ_SHADOW_FAST_CAP=27 timeout 900 python -m pytest tests/test_gpu_c4_shadow.py tests/test_gpu_parity.py -q -k "shadow or sw_dir" > $O/tests27.log 2>&1 ); tail -3 $O/tests27.log
( HZ_SHADOW_FAST_CAP=6 timeout 900 python -m pytest tests/test_gpu_c4_shadow.py tests/test_gpu_parity.py -q -k "shadow or sw_dir" > $O/tests6.log 2>&1 ); tail -3 $O/tests6.log
