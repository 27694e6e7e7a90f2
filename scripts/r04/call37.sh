#!/bin/bash
# round 4, GPU call 37: hand-rotated traversal loop (rot) against HEAD (new) and the first rewrite (c1); whole tile with HEAD
export TMPDIR=/tmp
O=gpurun_out/r04_37; mkdir -p $O
for round in 1 2 3; do
for v in c1 new rot; do
  if [ $v = new ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=$PWD/horayzon_amd/libhorayzon_hip_$v.so; fi
  ( timeout 120 python scripts/quick_perf.py --win 1024 --reps 3 > $O/q.tmp 2>&1 ); echo "$v $(grep 'rep 1\|rep 2' $O/q.tmp | awk '{print $6}' | tr '\n' ' ')" >> $O/ab.log
done
done
unset HORAYZON_HIP_LIB
cat $O/ab.log
( timeout 600 python bench.py --steps 3 --no-extras --no-e2e --no-cpu-baseline --no-peaks > $O/bench_tile.json 2>$O/bench_tile.err ); tail -1 $O/bench_tile.json | cut -c1-600
