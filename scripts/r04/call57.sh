#!/bin/bash
# round 4, GPU call 57: random sweep with the shadow kernel's retry path forced (8-entry fast stack), another seed
export TMPDIR=/tmp
O=gpurun_out/r04_57; mkdir -p $O
( HZ_SHADOW_FAST_CAP=8 HZ_FUZZ_N=150 HZ_FUZZ_SEED=47001 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -q -x -k "not stray" > $O/fuzz_47001_shadow_cap8.log 2>&1 ); tail -2 $O/fuzz_47001_shadow_cap8.log
