#!/bin/bash
# round 5, GPU call 24: persistent waves in k_horizon (every wave pulls 8 x 8 blocks from its XCD's queue) -- parity subset, then same-box
# A/B against one tile per workgroup (HZ_PERSIST=0): whole tile and a 1/8-tile slab
export TMPDIR=/tmp
O=gpurun_out/r05_24; mkdir -p $O
( time timeout 900 python -c "import torch; print(torch.__version__)" ) > $O/torch_import.log 2>&1
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q --durations=5 > $O/tests_parity.log 2>&1 ); tail -3 $O/tests_parity.log
for rep in 1 2; do
for pz in 0 1; do
  ( HZ_PERSIST=$pz timeout 300 python scripts/quick_perf.py --win 3569 --reps 2 > $O/whole_p${pz}_$rep.log 2>&1 ); echo whole persist $pz rep $rep $(grep "^rep" $O/whole_p${pz}_$rep.log | awk '{print $6}' | tr '\n' ' ')
  ( HZ_PERSIST=$pz timeout 300 python bench.py --rows-per-step 447 --steps 8 --warmup 3 --no-extras --no-e2e --no-cpu-baseline --no-count --no-peaks > $O/slab_p${pz}_$rep.json 2> $O/slab_p${pz}_$rep.err ); echo slab447 persist $pz rep $rep $(python -c "import json; d=json.loads(open('$O/slab_p${pz}_$rep.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms_per_launch'])" 2>&1 | tail -1)
done
done
