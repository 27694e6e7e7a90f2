#!/bin/bash
# round 5, GPU call 30: probe -DHZ_PROBE_DONE -- how much of a block's time do its lanes sit finished (cells done with all azimuths) while
# the wave waits for the block's slowest cells?
export TMPDIR=/tmp
O=gpurun_out/r05_30; mkdir -p $O
( time timeout 900 python -c "import torch; print(torch.__version__)" ) > $O/torch_import.log 2>&1
export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_done.so
( timeout 300 python scripts/quick_perf.py --win 3569 --reps 2 > $O/probe_done_whole.log 2>&1 ); grep "probe done\|^rep" $O/probe_done_whole.log
( timeout 300 python scripts/quick_perf.py --win 3569 --reps 1 --alg binary_search > $O/probe_done_binary.log 2>&1 ); grep "probe done\|^rep" $O/probe_done_binary.log
