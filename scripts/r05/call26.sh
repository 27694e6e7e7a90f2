#!/bin/bash
# round 5, GPU call 26: persistent waves after the register fixes (lane-derived values and block number out of the vector registers:
# node step back at 75 VALU instructions, no scratch) -- parity subset, A/B against HZ_PERSIST=0, compaction threshold re-swept under
# persistence, per-block start / end trace of a 1/8-tile slab (probe build -DHZ_WG_TRACE)
export TMPDIR=/tmp
O=gpurun_out/r05_26; mkdir -p $O
( time timeout 900 python -c "import torch; print(torch.__version__)" ) > $O/torch_import.log 2>&1
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q --durations=5 > $O/tests_parity.log 2>&1 ); tail -2 $O/tests_parity.log
for rep in 1 2; do
for pz in 0 1; do
  ( HZ_PERSIST=$pz timeout 300 python scripts/quick_perf.py --win 3569 --reps 2 > $O/whole_p${pz}_$rep.log 2>&1 ); echo whole persist $pz rep $rep $(grep "^rep" $O/whole_p${pz}_$rep.log | awk '{print $6}' | tr '\n' ' ')
  ( HZ_PERSIST=$pz timeout 300 python bench.py --rows-per-step 447 --steps 8 --warmup 3 --no-extras --no-e2e --no-cpu-baseline --no-count --no-peaks > $O/slab_p${pz}_$rep.json 2> $O/slab_p${pz}_$rep.err ); echo slab447 persist $pz rep $rep $(python -c "import json; d=json.loads(open('$O/slab_p${pz}_$rep.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms_per_launch'])" 2>&1 | tail -1)
done
done
for rg in 28 32 40 44; do
  ( timeout 300 python scripts/quick_perf.py --win 3569 --reps 2 --regroup $rg > $O/whole_rg$rg.log 2>&1 ); echo whole persist 1 regroup $rg $(grep "^rep" $O/whole_rg$rg.log | awk '{print $6}' | tr '\n' ' ')
done
for b in 20 28; do
  ( timeout 300 python scripts/quick_perf.py --win 3569 --reps 2 --regroup $((36 + b * 256)) > $O/whole_bias$b.log 2>&1 ); echo whole persist 1 regroup 36 bias $b $(grep "^rep" $O/whole_bias$b.log | awk '{print $6}' | tr '\n' ' ')
done
export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_trace.so
( HZ_WG_TRACE_OUT=$O/trace_slab447.txt timeout 300 python bench.py --rows-per-step 447 --steps 1 --warmup 1 --no-extras --no-e2e --no-cpu-baseline --no-count --no-peaks > $O/trace_bench.json 2> $O/trace_bench.err )
python scripts/wg_trace_stats.py $O/trace_slab447.txt $O/wg_trace_persistent_slab447.json > /dev/null 2>&1; python -c "
import json; d=json.load(open('$O/wg_trace_persistent_slab447.json')); x=d[-1]; print({k:x[k] for k in ('launch','end_ms','last_wave_started_ms','tail_ms_after_last_start','max_resident_waves','efficiency_vs_full_residency','resident_waves_at_fraction_of_launch')}); print(x['wave_lifetime_ms'])"
rm -f $O/trace_slab447.txt
