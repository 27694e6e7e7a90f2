#!/bin/bash
# round 6, GPU call 18: does the targeted test catch the hole?  The same tree with hz_near.hip of commit 85536e0 (before rows F' / C' / B')
O=gpurun_out/r06_18
mkdir -p $O
HORAYZON_HIP_LIB=$PWD/horayzon_amd/libhorayzon_hip_prefix_near.so timeout 600 python -m pytest tests/test_gpu_near_guard.py -q -m gpu -k "spike" 2>&1 | grep -E "^E  |passed|failed" | cut -c1-400 | head -12 > $O/near_guard_with_the_old_prepass.log
cat $O/near_guard_with_the_old_prepass.log
