#!/bin/bash
# round 5, GPU call 35: the default bench line with the refreshed counter profiles (valu model, traffic) of the final kernel sources
export TMPDIR=/tmp
O=gpurun_out/r05_35; mkdir -p $O
( time timeout 900 python -c "import torch; print(torch.__version__)" ) > $O/torch_import.log 2>&1
( timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_line.err ); tail -1 $O/bench_line.json | cut -c1-400
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 ); tail -1 $O/smoke.log
