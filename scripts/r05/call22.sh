#!/bin/bash
# round 5, GPU call 22: k_near_cert with the polar form per vertex and the in-plane candidates per vertex (instead of per edge end and bin)
# against the committed kernel (libhorayzon_hip_nearold.so): certificate tests, pre-pass time on the whole tile, and an adversarial sweep run
# with BOTH libraries on the same seed -- the per-configuration `shortened` counts must be the same (same certificates), 0 violations
export TMPDIR=/tmp
O=gpurun_out/r05_22; mkdir -p $O
( time timeout 600 python -c "import torch; print(torch.__version__)" ) > $O/torch_import.log 2>&1
( timeout 1500 python -m pytest tests/test_gpu_near_guard.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q > $O/tests_near.log 2>&1 ); tail -3 $O/tests_near.log
for rep in 1 2; do
for lib in nearold product; do
  if [ $lib = product ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_$lib.so; fi
  ( timeout 300 python scripts/quick_perf.py --win 3569 --reps 3 > $O/qp_${lib}_$rep.log 2>&1 ); echo qp $lib $rep $(grep "^rep 2" $O/qp_${lib}_$rep.log | cut -c1-60) $(grep "near pre-pass" $O/qp_${lib}_$rep.log | tail -1)
done
done
unset HORAYZON_HIP_LIB
( timeout 1200 python scripts/fuzz_near_adversarial.py --n 800 --seed 53001 --oracle-every 4 --out $O/fuzz_near_53001_new.jsonl 2> $O/fuzz_new.err ); tail -1 $O/fuzz_near_53001_new.jsonl | cut -c1-400
( HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_nearold.so timeout 1200 python scripts/fuzz_near_adversarial.py --n 800 --seed 53001 --oracle-every 0 --out $O/fuzz_near_53001_old.jsonl 2> $O/fuzz_old.err ); tail -1 $O/fuzz_near_53001_old.jsonl | cut -c1-400
python - <<'PY'
import json
O = "gpurun_out/r05_22/"
a = [json.loads(l) for l in open(O + "fuzz_near_53001_new.jsonl") if l.startswith("{")]
b = [json.loads(l) for l in open(O + "fuzz_near_53001_old.jsonl") if l.startswith("{")]
a = [r for r in a if "i" in r]; b = [r for r in b if "i" in r]
diff = [(x["i"], x["shortened"], y["shortened"]) for x, y in zip(a, b) if (x["shortened"], x["rays"], x["near_used"]) != (y["shortened"], y["rays"], y["near_used"])]
print("configs compared", len(a), len(b), "differing (i, new, old):", diff[:20], "n_diff", len(diff))
print("violations new", sum(r["violations"] for r in a), "old", sum(r["violations"] for r in b), "shortened new", sum(r["shortened"] for r in a), "old", sum(r["shortened"] for r in b))
PY
