#!/bin/bash
# round 4, GPU call 41: rocprofv3 passes on the final kernels (r04s), then the lines of record
export TMPDIR=/tmp
O=gpurun_out/r04_41; mkdir -p $O
bash scripts/profile_bench.sh r04s > $O/profile.log 2>&1; tail -1 $O/profile.log | cut -c1-200
