#!/bin/bash
# round 5, GPU call 28: persistent waves in k_shadow_refill, second attempt (a plain loop around the kernel body as in k_horizon: one
# item after the other; sun position in scalar registers) -- shadow parity, then config 4 against HZ_PERSIST=0 with 32 / 16 / 8 blocks per item
export TMPDIR=/tmp
O=gpurun_out/r05_28; mkdir -p $O
( time timeout 900 python -c "import torch; print(torch.__version__)" ) > $O/torch_import.log 2>&1
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_c4_shadow.py -m gpu -x -q --durations=5 -k "shadow or persistent or terrain" > $O/tests_shadow.log 2>&1 ); tail -2 $O/tests_shadow.log
for rep in 1 2; do
for cfg in "0 0" "1 0" "1 16" "1 8"; do
  set -- $cfg
  export HZ_PERSIST=$1
  if [ $2 = 0 ]; then unset HZ_SHADOW_NB; else export HZ_SHADOW_NB=$2; fi
  for rf in 0 1; do
  ( timeout 300 python bench.py --workload c4 --refrac $rf > $O/c4_p$1_nb$2_rf${rf}_$rep.json 2> $O/c4_p$1_nb$2_rf${rf}_$rep.err ); echo c4 persist $1 nb $2 refrac $rf rep $rep $(python -c "import json; d=json.loads(open('$O/c4_p$1_nb$2_rf${rf}_$rep.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)
  done
done
done
