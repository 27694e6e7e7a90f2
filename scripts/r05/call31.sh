#!/bin/bash
# round 5, GPU call 31: leftover cells -- a block ends when <= HZ_LEFT_MIN of its cells are unfinished, a second launch finishes them.
# Parity (small grids hand over too), then whole tile / slab with HZ_LEFT_MIN = 0 (off) / 8 / 16 / 24 / 32
export TMPDIR=/tmp
O=gpurun_out/r05_31; mkdir -p $O
( time timeout 900 python -c "import torch; print(torch.__version__)" ) > $O/torch_import.log 2>&1
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_near_guard.py -m gpu -x -q --durations=5 > $O/tests_parity.log 2>&1 ); tail -3 $O/tests_parity.log
for lm in 0 16 8 24 32 0 16; do
  ( HZ_LEFT_TRACE=1 HZ_LEFT_MIN=$lm timeout 300 python scripts/quick_perf.py --win 3569 --reps 2 > $O/whole_lm${lm}.log 2>&1 ); echo whole left_min $lm $(grep "^rep" $O/whole_lm${lm}.log | awk '{print $6}' | tr '\n' ' ') $(grep "leftover cells" $O/whole_lm${lm}.log | tail -1)
done
for lm in 0 16; do
  ( HZ_LEFT_MIN=$lm timeout 300 python bench.py --rows-per-step 447 --steps 8 --warmup 3 --no-extras --no-e2e --no-cpu-baseline --no-count --no-peaks > $O/slab_lm${lm}.json 2> $O/slab_lm${lm}.err ); echo slab447 left_min $lm $(python -c "import json; d=json.loads(open('$O/slab_lm${lm}.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms_per_launch'])" 2>&1 | tail -1)
done
