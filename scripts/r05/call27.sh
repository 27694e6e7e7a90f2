#!/bin/bash
# round 5, GPU call 27: experiment HZ_PERSIST=2 (a workgroup pulls whole tiles and hands the quadrants to its waves through LDS) against
# HZ_PERSIST=1 (every wave pulls blocks on its own), same library, same box
export TMPDIR=/tmp
O=gpurun_out/r05_27; mkdir -p $O
( time timeout 900 python -c "import torch; print(torch.__version__)" ) > $O/torch_import.log 2>&1
( HZ_PERSIST=2 HZ_PERSIST_GRID=3 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c2_gaussian_hill or rough_tilted or mask_and_fill or row_slab" > $O/tests_p2_small.log 2>&1 ); tail -2 $O/tests_p2_small.log
( HZ_PERSIST=2 timeout 400 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q > $O/tests_p2_full.log 2>&1 ); tail -2 $O/tests_p2_full.log
for rep in 1 2 3; do
for pz in 1 2; do
  ( HZ_PERSIST=$pz timeout 200 python scripts/quick_perf.py --win 3569 --reps 2 > $O/whole_p${pz}_$rep.log 2>&1 ); echo whole persist $pz rep $rep $(grep "^rep" $O/whole_p${pz}_$rep.log | awk '{print $6}' | tr '\n' ' ')
done
done
for pz in 1 2; do
  ( HZ_PERSIST=$pz timeout 300 python bench.py --rows-per-step 447 --steps 8 --warmup 3 --no-extras --no-e2e --no-cpu-baseline --no-count --no-peaks > $O/slab_p${pz}.json 2> $O/slab_p${pz}.err ); echo slab447 persist $pz $(python -c "import json; d=json.loads(open('$O/slab_p${pz}.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms_per_launch'])" 2>&1 | tail -1)
done
