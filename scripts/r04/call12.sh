#!/bin/bash
# round 4, GPU call 12: hit-cache level variants; balance test on the lowland/relief DEM; rocprofv3 passes over bench.py (final kernels)
export TMPDIR=/tmp
O=gpurun_out/r04_12; mkdir -p $O
for v in base anc4 anc6 base anc4 anc6; do
  if [ $v = base ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=$PWD/horayzon_amd/libhorayzon_hip_$v.so; fi
  ( timeout 300 python scripts/quick_perf.py --win 1024 --reps 3 > $O/q.tmp 2>&1 ); echo "$v $(grep 'rep 1\|rep 2' $O/q.tmp | awk '{print $6}' | tr '\n' ' ')" >> $O/anc_levels.log
done
unset HORAYZON_HIP_LIB
cat $O/anc_levels.log
( timeout 900 python -m pytest tests/test_gpu_bench_ranks.py -q -s -k "inhomogeneous" > $O/tests.log 2>&1 ); tail -3 $O/tests.log; grep '"cost"' $O/tests.log | cut -c1-900
bash scripts/profile_bench.sh r04p > $O/profile.log 2>&1; tail -3 $O/profile.log
