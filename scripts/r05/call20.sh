#!/bin/bash
# round 5, GPU call 20: after the cache walks stay in the loop -- (a) compaction threshold / vote bias re-swept (opts.regroup = threshold | bias << 8),
# (b) hit-cache ancestor level 4 / 5 / 6, (c) hz_crmath.h coefficients pinned to scalar registers at their step (-DHZ_CRM_PIN_SGPR: 1 instead
# of 29 spilled VGPRs in k_shadow_refill, arithmetic unchanged) and the same with fused steps (timing only): config 4 +- refraction
export TMPDIR=/tmp
O=gpurun_out/r05_20; mkdir -p $O
( time timeout 600 python -c "import torch; print(torch.__version__)" ) > $O/torch_import.log 2>&1
for rg in 6180 6176 6184 6172 6188 5156 7204 6180; do
  ( timeout 300 python scripts/quick_perf.py --win 3569 --reps 2 --regroup $rg > $O/qp_regroup_$rg.log 2>&1 ); echo qp regroup $((rg & 255)) bias $((rg >> 8)) $(grep "^rep 1" $O/qp_regroup_$rg.log | cut -c1-120)
done
for lib in anc4 anc6 product anc4 anc6 product; do
  if [ $lib = product ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_$lib.so; fi
  ( timeout 300 python scripts/quick_perf.py --win 3569 --reps 2 > $O/qp_${lib}.log 2>&1 ); echo qp $lib $(grep "^rep 1" $O/qp_${lib}.log | cut -c1-120)
done
for rep in 1 2; do
for lib in pin fpin product; do
for rf in 1 0; do
  if [ $lib = product ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_$lib.so; fi
  ( timeout 300 python bench.py --workload c4 --refrac $rf > $O/c4_${lib}_rf${rf}_$rep.json 2> $O/c4_${lib}_rf${rf}_$rep.err ); echo c4 $lib refrac $rf rep $rep $(python -c "import json; d=json.loads(open('$O/c4_${lib}_rf${rf}_$rep.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)
done
done
done
