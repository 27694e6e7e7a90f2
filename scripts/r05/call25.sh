#!/bin/bash
# round 5, GPU call 25: persistent waves in k_shadow_refill (items pulled per wave, lanes fed across items) -- shadow parity, then
# config 4 against the library of commit f454b08 (libhorayzon_hip_r5a.so) with 32 / 16 / 8 / 4 blocks per item
export TMPDIR=/tmp
O=gpurun_out/r05_25; mkdir -p $O
( time timeout 900 python -c "import torch; print(torch.__version__)" ) > $O/torch_import.log 2>&1
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_c4_shadow.py -m gpu -x -q --durations=5 -k "shadow or persistent or terrain" > $O/tests_shadow.log 2>&1 ); tail -3 $O/tests_shadow.log
for rep in 1 2; do
for cfg in "r5a 0" "new 0" "new 16" "new 8" "new 4"; do
  set -- $cfg
  if [ $1 = new ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_$1.so; fi
  if [ $2 = 0 ]; then unset HZ_SHADOW_NB; else export HZ_SHADOW_NB=$2; fi
  for rf in 0 1; do
  ( timeout 300 python bench.py --workload c4 --refrac $rf > $O/c4_$1_nb$2_rf${rf}_$rep.json 2> $O/c4_$1_nb$2_rf${rf}_$rep.err ); echo c4 $1 nb $2 refrac $rf rep $rep $(python -c "import json; d=json.loads(open('$O/c4_$1_nb$2_rf${rf}_$rep.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)
  done
done
done
