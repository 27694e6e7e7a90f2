#!/bin/bash
# round 5, GPU call 15: the tree that is meant to be final -- whole GPU suite, the default bench line, locations threshold 16 / 12 / 8,
# then the rocprofv3 passes (kernel trace + stats, FETCH_SIZE / WRITE_SIZE / SQ counters in their own PMC passes) for profiles/r05
export TMPDIR=/tmp
O=gpurun_out/r05_15; mkdir -p $O
( time timeout 600 python -c "import torch; print(torch.__version__)" ) > $O/torch_import.log 2>&1
( timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 > $O/tests_gpu.log 2>&1 ); tail -12 $O/tests_gpu.log
for lib in product lr12 lr8; do
  if [ $lib = product ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_$lib.so; fi
  ( timeout 300 python scripts/quick_locations.py --reps 3 > $O/loc_$lib.log 2>&1 ); echo locations $lib $(grep "^rep" $O/loc_$lib.log | sed 's/.*kernel \([0-9.]*\)s.*/\1/' | tr '\n' ' ')
done
unset HORAYZON_HIP_LIB
( time timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_line.err ); tail -c 600 $O/bench_line.json; tail -3 $O/bench_line.err
bash scripts/profile_bench.sh r05a > $O/profile.log 2>&1; tail -2 $O/profile.log | cut -c1-300
ls gpurun_out | head -40
