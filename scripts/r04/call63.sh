#!/bin/bash
# round 4, GPU call 63: smoke() and the oracle-comparing quick tests after the oracle got its experiment hook
export TMPDIR=/tmp
O=gpurun_out/r04_63; mkdir -p $O
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 ); tail -1 $O/smoke.log
( timeout 600 python -m pytest tests/test_gpu_parity.py -q -x > $O/parity.log 2>&1 ); tail -1 $O/parity.log
