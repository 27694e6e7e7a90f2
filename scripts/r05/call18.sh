#!/bin/bash
# round 5, GPU call 18 (re-entry after the container was replaced; call 17's outputs were lost): fused Horner steps in hz_crmath.h:
# config 4 A/B against the unfused build; the FULL GPU suite on the final tree; plain `python bench.py --gpus 2` with both ranks on
# the one GPU (gloo)
export TMPDIR=/tmp
O=gpurun_out/r05_18; mkdir -p $O
( time timeout 600 python -c "import torch; print(torch.__version__)" ) > $O/torch_import.log 2>&1
( timeout 2400 python -m pytest tests -m gpu -x -q --durations=10 > $O/tests_gpu_full.log 2>&1 ); tail -4 $O/tests_gpu_full.log
for rep in 1 2 3; do
for lib in crmold product; do
  if [ $lib = product ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_$lib.so; fi
  ( timeout 300 python bench.py --workload c4 --refrac 1 > $O/c4_${lib}_$rep.json 2> $O/c4_${lib}_$rep.err ); echo c4 $lib refrac 1 rep $rep $(python -c "import json; d=json.loads(open('$O/c4_${lib}_$rep.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)
done
done
unset HORAYZON_HIP_LIB
( HZ_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --no-cpu-baseline > $O/bench_gpus2_gloo.json 2> $O/bench_gpus2_gloo.err ); tail -c 3000 $O/bench_gpus2_gloo.json | head -c 2500; echo; tail -2 $O/bench_gpus2_gloo.err
