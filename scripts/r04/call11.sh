#!/bin/bash
# round 4, GPU call 11: shadow leaf-bias variants; whole suite; cost balance with refined probes
export TMPDIR=/tmp
O=gpurun_out/r04_11; mkdir -p $O
rm -f gpurun_out/r04_near_verify.jsonl
for v in base sb24 sb28 base sb24 sb28; do
  if [ $v = base ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=$PWD/horayzon_amd/libhorayzon_hip_$v.so; fi
  ( timeout 300 python bench.py --workload c4 --no-count --no-peaks > $O/c4.tmp 2>/dev/null ); echo "$v $(tail -1 $O/c4.tmp | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["ms_per_sun_position"])')" >> $O/shadow_bias.log
  ( timeout 300 python bench.py --workload c4 --refrac 1 --which sw_dir_cor --no-count --no-peaks > $O/c4.tmp 2>/dev/null ); echo "$v refrac sw_dir_cor $(tail -1 $O/c4.tmp | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["ms_per_sun_position"])')" >> $O/shadow_bias.log
done
unset HORAYZON_HIP_LIB
cat $O/shadow_bias.log
( timeout 2700 python -m pytest tests -m gpu -q > $O/tests_gpu.log 2>&1 ); tail -6 $O/tests_gpu.log
( timeout 300 python scripts/quick_perf.py --win 1024 --reps 4 --count > $O/quick.log 2>&1 ); grep "rep\|SIMT" $O/quick.log
cp gpurun_out/r04_near_verify.jsonl $O/ 2>/dev/null
