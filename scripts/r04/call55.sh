#!/bin/bash
# round 4, GPU call 55: the new shadow retry test
export TMPDIR=/tmp
O=gpurun_out/r04_55; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_c4_shadow.py -q -k "overflowed" > $O/t.log 2>&1 ); tail -3 $O/t.log
