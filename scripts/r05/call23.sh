#!/bin/bash
# round 5, GPU call 23: the FULL GPU suite on the tree after the second half of the round (cache walks in the loop, ancestor level 4, pinned
# coefficients, constant divisions, per-vertex k_near_cert), the counter-example replay, 300 random configurations against the oracle, and
# the shadow kernel at 8 workgroups per CU (13 spilled VGPRs now; 18 / 17 stack entries so that 8 fit) against the product's 7
export TMPDIR=/tmp
O=gpurun_out/r05_23; mkdir -p $O
( time timeout 600 python -c "import torch; print(torch.__version__)" ) > $O/torch_import.log 2>&1
( timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > $O/tests_gpu_full.log 2>&1 ); tail -4 $O/tests_gpu_full.log
( timeout 300 python scripts/replay_adv.py 48001 2536 > $O/replay_adv_48001_2536.log 2>&1 ); cat $O/replay_adv_48001_2536.log | cut -c1-250
( HZ_FUZZ_N=300 HZ_FUZZ_SEED=53003 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -x -k "not stray" > $O/fuzz_53003.log 2>&1 ); tail -3 $O/fuzz_53003.log
for rep in 1 2; do
for cfg in "product 19" "sh8 18" "sh8 17"; do
  set -- $cfg
  if [ $1 = product ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_$1.so; fi
  for rf in 0 1; do
  ( HZ_SHADOW_FAST_CAP=$2 timeout 300 python bench.py --workload c4 --refrac $rf > $O/c4_$1_$2_rf${rf}_$rep.json 2> $O/c4_$1_$2_rf${rf}_$rep.err ); echo c4 $1 cap $2 refrac $rf rep $rep $(python -c "import json; d=json.loads(open('$O/c4_$1_$2_rf${rf}_$rep.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)
  done
done
done
