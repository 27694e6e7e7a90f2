#!/bin/bash
# round 5, GPU call 5: the whole GPU suite after the per-cell guard / dead-node / box-start changes (torch imported once
# up front and timed: on a fresh box the first import pages the image in), then slab-launch probes: 1/8-tile slabs with
# 4-, 2- and 1-wave workgroups and a per-wave start / end trace of one slab
export TMPDIR=/tmp
O=gpurun_out/r05_05; mkdir -p $O
( time timeout 600 python -c "import torch; print(torch.__version__, torch.cuda.is_available())" ) > $O/torch_import.log 2>&1; tail -4 $O/torch_import.log
( timeout 1800 python -m pytest tests -m gpu -x -q --durations=12 > $O/tests_gpu.log 2>&1 ); tail -16 $O/tests_gpu.log
for rep in 1 2; do
  for lib in product wpb2 wpb1; do
    if [ $lib = product ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_$lib.so; fi
    ( timeout 300 python bench.py --rows-per-step 446 --steps 8 --warmup 1 --no-cpu-baseline --no-count --no-peaks --no-e2e --no-extras > $O/slab_${lib}_$rep.json 2> $O/slab_${lib}_$rep.err )
    echo slab446 $lib $rep $(python -c "import json,sys; d=json.loads(open('$O/slab_${lib}_$rep.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline'].get('kernel_ms_per_launch'))")
    ( timeout 300 python scripts/quick_perf.py --win 3569 --reps 3 > $O/perf_${lib}_$rep.log 2>&1 ); echo whole $lib $rep $(grep "^rep" $O/perf_${lib}_$rep.log | sed 's/.*kernel \([0-9.]*\)s.*/\1/' | tr '\n' ' ')
  done
done
export HZ_WG_TRACE_OUT=$PWD/$O/wg_trace_wpb4.txt
HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_trace.so timeout 300 python bench.py --rows-per-step 446 --steps 2 --warmup 0 --no-cpu-baseline --no-count --no-peaks --no-e2e --no-extras > /dev/null 2> $O/trace4.err
export HZ_WG_TRACE_OUT=$PWD/$O/wg_trace_wpb1.txt
HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_wpb1trace.so timeout 300 python bench.py --rows-per-step 446 --steps 2 --warmup 0 --no-cpu-baseline --no-count --no-peaks --no-e2e --no-extras > /dev/null 2> $O/trace1.err
unset HORAYZON_HIP_LIB HZ_WG_TRACE_OUT
ls -la $O/*.txt; for f in $O/wg_trace_wpb4.txt $O/wg_trace_wpb1.txt; do gzip -f $f; done
