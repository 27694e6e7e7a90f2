#!/bin/bash
# round 4, GPU call 58: certificate pre-pass with (sin, cos) pairs and a stepped azimuth index in its inner loop
export TMPDIR=/tmp
O=gpurun_out/r04_58; mkdir -p $O
for round in 1 2; do
for v in nearprev new; do
  if [ $v = new ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=$PWD/horayzon_amd/libhorayzon_hip_$v.so; fi
  ( timeout 300 python bench.py --steps 3 --no-extras --no-e2e --no-cpu-baseline --no-count --no-peaks > $O/b.tmp 2>/dev/null ); echo "$v near_prepass_ms $(tail -1 $O/b.tmp | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["near_prepass_ms_per_step"], d["ms_per_step"])')" >> $O/ab_near.log
done
done
unset HORAYZON_HIP_LIB
cat $O/ab_near.log
( timeout 1200 python -m pytest tests/test_gpu_near_guard.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -q -k "near or certificate or adversarial or retraced" > $O/tests.log 2>&1 ); tail -3 $O/tests.log
( timeout 900 python scripts/fuzz_near_adversarial.py --n 2600 --seed 48001 --oracle-every 4 --out $O/fuzz_near_48001.jsonl 2> $O/fuzz_near_48001.err ); tail -1 $O/fuzz_near_48001.jsonl | cut -c1-300
