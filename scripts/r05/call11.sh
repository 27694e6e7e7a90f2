#!/bin/bash
# round 5, GPU call 11: (a) how many entries does the fast stack need on the C3 tile?  (a leaf-pooling experiment would pay for its
# LDS with stack entries: 27 -> ~20); (b) the shadow kernel's compaction threshold re-swept (40 / 36 / 32 / 28), config 4
export TMPDIR=/tmp
O=gpurun_out/r05_11; mkdir -p $O
for n in 27 22 20 18 16; do
  ( timeout 300 python scripts/quick_perf.py --win 3569 --reps 2 --stack $n > $O/perf_stack$n.log 2>&1 ); echo stack $n $(grep "^rep" $O/perf_stack$n.log | sed 's/.*kernel \([0-9.]*\)s.*/\1/' | tr '\n' ' ') $(grep "redo" $O/perf_stack$n.log | tail -1)
done
for lib in product sr36 sr32 sr28; do
  if [ $lib = product ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_$lib.so; fi
  for rf in 0 1; do
    ( timeout 300 python bench.py --workload c4 --refrac $rf > $O/c4_${lib}_refrac$rf.json 2> $O/c4_${lib}_refrac$rf.err ); echo c4 $lib refrac $rf $(python -c "import json; d=json.loads(open('$O/c4_${lib}_refrac$rf.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)
  done
done
