#!/bin/bash
# round 5, GPU call 7 (the container was re-created after call 6 was sent: its outputs were lost): the WHOLE GPU suite on the
# tree of 31dadbc, the counter-example replay, 1/8-tile slabs r4 / product / 2- and 1-wave workgroups on eight equal slabs
# (447 rows, 8 + 8 steps), per-wave traces of a slab, config 4 with and without refraction against r4, whole-tile A/B
# against r4, and the default bench line
export TMPDIR=/tmp
O=gpurun_out/r05_07; mkdir -p $O
( time timeout 600 python -c "import torch; print(torch.__version__)" ) > $O/torch_import.log 2>&1
( timeout 1800 python -m pytest tests -m gpu -x -q --durations=12 > $O/tests_gpu.log 2>&1 ); tail -16 $O/tests_gpu.log
( timeout 300 python scripts/replay_adv.py 48001 2536 > $O/replay_adv_48001_2536.log 2>&1 ); cat $O/replay_adv_48001_2536.log
for rep in 1 2; do
  for lib in r4 product wpb2 wpb1; do
    if [ $lib = product ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_$lib.so; fi
    ( timeout 300 python bench.py --rows-per-step 447 --steps 8 --warmup 8 --no-cpu-baseline --no-count --no-peaks --no-e2e --no-extras > $O/slab_${lib}_$rep.json 2> $O/slab_${lib}_$rep.err )
    echo slab447 $lib $rep $(python -c "import json,sys; d=json.loads(open('$O/slab_${lib}_$rep.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline'].get('kernel_ms_per_launch'))" 2>&1 | tail -1)
    ( timeout 300 python scripts/quick_perf.py --win 3569 --reps 3 > $O/perf_${lib}_$rep.log 2>&1 ); echo whole $lib $rep $(grep "^rep" $O/perf_${lib}_$rep.log | sed 's/.*kernel \([0-9.]*\)s.*/\1/' | tr '\n' ' ')
  done
done
export HZ_WG_TRACE_OUT=$PWD/$O/wg_trace_wpb4.txt
HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_trace.so timeout 300 python bench.py --rows-per-step 447 --steps 2 --warmup 0 --no-cpu-baseline --no-count --no-peaks --no-e2e --no-extras > $O/trace4.out 2> $O/trace4.err; tail -3 $O/trace4.err
export HZ_WG_TRACE_OUT=$PWD/$O/wg_trace_wpb1.txt
HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_wpb1trace.so timeout 300 python bench.py --rows-per-step 447 --steps 2 --warmup 0 --no-cpu-baseline --no-count --no-peaks --no-e2e --no-extras > $O/trace1.out 2> $O/trace1.err; tail -3 $O/trace1.err
unset HORAYZON_HIP_LIB HZ_WG_TRACE_OUT
ls -la $O/*.txt; for f in $O/wg_trace_wpb4.txt $O/wg_trace_wpb1.txt; do gzip -f $f; done
for rf in 0 1; do
  ( timeout 300 python bench.py --workload c4 --refrac $rf > $O/c4_refrac$rf.json 2> $O/c4_refrac$rf.err ); echo c4 refrac $rf $(python -c "import json; d=json.loads(open('$O/c4_refrac$rf.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)
  HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_r4.so timeout 300 python bench.py --workload c4 --refrac $rf > $O/c4_r4_refrac$rf.json 2> /dev/null; echo c4 r4 refrac $rf $(python -c "import json; d=json.loads(open('$O/c4_r4_refrac$rf.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)
done
( time timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_line.err ); tail -c 1500 $O/bench_line.json; tail -3 $O/bench_line.err
