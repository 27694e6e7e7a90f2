#!/bin/bash
# round 5, GPU call 8: the build-time switch -DHZ_TRI_FMA (cross / dot products of the triangle test with FMAs, as Embree's vector
# code evaluates them): bit parity against the oracle's "plain_fma" mode, then what it is worth -- whole C3 tile product / fma,
# the leaf-vote bias re-swept for the cheaper leaf step, config 4, a 1/8-tile slab
export TMPDIR=/tmp
O=gpurun_out/r05_08; mkdir -p $O
( time timeout 600 python -c "import torch; print(torch.__version__)" ) > $O/torch_import.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_tri_fma.py -x -q -s > $O/test_tri_fma.log 2>&1 ); tail -5 $O/test_tri_fma.log
for rep in 1 2; do
  for lib in product fma; do
    if [ $lib = product ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_$lib.so; fi
    ( timeout 300 python scripts/quick_perf.py --win 3569 --reps 3 --count > $O/perf_${lib}_$rep.log 2>&1 ); echo whole $lib $rep $(grep "^rep" $O/perf_${lib}_$rep.log | sed 's/.*kernel \([0-9.]*\)s.*/\1/' | tr '\n' ' ') $(grep SIMT $O/perf_${lib}_$rep.log)
  done
done
export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_fma.so
for rg in 4136 5160 7208 8232 6192 6176; do
  ( timeout 300 python scripts/quick_perf.py --win 3569 --reps 2 --regroup $rg > $O/perf_fma_rg$rg.log 2>&1 ); echo fma regroup $rg thr $((rg & 255)) bias $((rg >> 8)) $(grep "^rep" $O/perf_fma_rg$rg.log | sed 's/.*kernel \([0-9.]*\)s.*/\1/' | tr '\n' ' ')
done
( timeout 300 python bench.py --rows-per-step 447 --steps 8 --warmup 8 --no-cpu-baseline --no-count --no-peaks --no-e2e --no-extras > $O/slab_fma.json 2> $O/slab_fma.err )
echo slab447 fma $(python -c "import json,sys; d=json.loads(open('$O/slab_fma.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline'].get('kernel_ms_per_launch'))" 2>&1 | tail -1)
for rf in 0 1; do
  ( timeout 300 python bench.py --workload c4 --refrac $rf > $O/c4_fma_refrac$rf.json 2> $O/c4_fma_refrac$rf.err ); echo c4 fma refrac $rf $(python -c "import json; d=json.loads(open('$O/c4_fma_refrac$rf.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)
done
unset HORAYZON_HIP_LIB
for rf in 0 1; do
  ( timeout 300 python bench.py --workload c4 --refrac $rf > $O/c4_refrac$rf.json 2> $O/c4_refrac$rf.err ); echo c4 product refrac $rf $(python -c "import json; d=json.loads(open('$O/c4_refrac$rf.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" 2>&1 | tail -1)
done
