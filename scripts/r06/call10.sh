#!/bin/bash
# round 6, GPU call 10: diagnosis of the two certificate violations of sweep seed 64003
mkdir -p gpurun_out/r06_10
HORAYZON_VERBOSE=0 timeout 600 python scripts/r06/diag_64003.py > gpurun_out/r06_10/diag.log 2>&1
cat gpurun_out/r06_10/diag.log | cut -c1-900
