#!/bin/bash
# round 4, GPU call 4: certificate pre-pass v2 (single evaluation + exact interval bound, reasons), VERIFY without extra VGPRs,
# HostPinner rewrite, persistent sun buffer
export TMPDIR=/tmp
O=gpurun_out/r04_4; mkdir -p $O
rm -f gpurun_out/r04_near_verify.jsonl
( timeout 300 python scripts/r04/near_reasons_diag.py > /dev/null 2> $O/near_reasons.log ); grep -c CASE $O/near_reasons.log
( timeout 300 python scripts/quick_perf.py --win 1024 --reps 3 > $O/quick.log 2>&1 ); grep "rep\|near" $O/quick.log
( timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "certificates" > $O/tests_full.log 2>&1 ); tail -3 $O/tests_full.log
( timeout 1800 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py::test_c3_whole_tile_certificates_retraced --deselect tests/test_gpu_fullsize.py::test_c3_curved_tile_certificates_retraced > $O/tests_gpu.log 2>&1 ); tail -5 $O/tests_gpu.log
cp gpurun_out/r04_near_verify.jsonl $O/ 2>/dev/null
