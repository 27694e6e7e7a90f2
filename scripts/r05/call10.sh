#!/bin/bash
# round 5, GPU call 10: probe build -DHZ_PROBE_Q1 (counting instantiation): at leaf steps, how many lanes hold a SECOND queued
# leaf; at node steps, how many lanes are blocked by a full queue / have a decided ray (what would pooling the queued leaves of
# a wave over its idle lanes be worth?)
export TMPDIR=/tmp
O=gpurun_out/r05_10; mkdir -p $O
export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_q1probe.so
for rg in 6184 6180 6176; do
( timeout 300 python scripts/quick_perf.py --win 3569 --reps 2 --count --regroup $rg > $O/probe_q1_$rg.log 2>&1 ); grep "hz probe\|SIMT\|^rep" $O/probe_q1_$rg.log | cut -c1-400
done
