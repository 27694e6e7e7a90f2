#!/bin/bash
# round 4, GPU call 26: PC sampling of k_horizon on the 1024^2 probe (where do the waves wait?).  Every rocprofv3 run is
# bounded by its own timeout; only the aggregated histograms leave the box.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04_26; mkdir -p $O
cd /tmp
( timeout 120 rocprofv3 --list-avail > $O/avail.txt 2>&1 ); grep -n -i -A12 "pc sampl\|pc_sampl" $O/avail.txt | head -60
for M in stochastic host_trap; do
  if [ $M = stochastic ]; then U=cycles; I=1048576; else U=time; I=5000; fi
  rm -rf /tmp/pcs_$M
  ( timeout 420 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $M --pc-sampling-unit $U --pc-sampling-interval $I \
      --output-format csv -d /tmp/pcs_$M -- python $R/scripts/quick_perf.py --win 1024 --reps 3 > $O/pcs_$M.log 2>&1 )
  echo "$M exit $?"; tail -3 $O/pcs_$M.log | cut -c1-300
  python $R/scripts/pc_sample_hist.py /tmp/pcs_$M $O/pcs_$M 2>&1 | tail -5
done
rocm-smi --showuse 2>&1 | tail -5
