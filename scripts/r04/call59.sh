#!/bin/bash
# round 4, GPU call 60: replay of adversarial configuration 2536 of seed 48001 (oracle mismatch) with both libraries
export TMPDIR=/tmp
O=gpurun_out/r04_60; mkdir -p $O
for v in old new; do
  if [ $v = new ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=$PWD/horayzon_amd/libhorayzon_hip_$v.so; fi
  ( timeout 300 python scripts/r04/replay_adv.py 48001 2536 >> $O/replay.log 2>&1 )
done
grep -v amdgpu.ids $O/replay.log
