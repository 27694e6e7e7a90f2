#!/bin/bash
# round 5, GPU call 14: shadow compaction threshold 20 / 16 / 12 / 8, locations kernel threshold 32 / 28 / 24 / 16
export TMPDIR=/tmp
O=gpurun_out/r05_14; mkdir -p $O
for rep in 1 2; do
for lib in sr20 sr16 sr12 sr8; do
  export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_$lib.so
  for rf in 0 1; do
    ( timeout 300 python bench.py --workload c4 --refrac $rf > $O/c4_${lib}_refrac${rf}_$rep.json 2> $O/c4_${lib}_refrac${rf}_$rep.err ); echo c4 $lib refrac $rf rep $rep $(python -c "import json; d=json.loads(open('$O/c4_${lib}_refrac${rf}_$rep.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)
  done
done
done
for lib in lr32 lr28 lr24 lr16; do
  export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_$lib.so
  ( timeout 300 python scripts/quick_locations.py --reps 3 > $O/loc_$lib.log 2>&1 ); echo locations $lib $(grep "^rep" $O/loc_$lib.log | sed 's/.*kernel \([0-9.]*\)s.*/\1/' | tr '\n' ' ')
done
