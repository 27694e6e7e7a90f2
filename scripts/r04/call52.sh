#!/bin/bash
# round 4, GPU call 52: wide sweeps on the final code: adversarial certificates (2 x 2600, every second one against the oracle),
# random configurations (300, another seed)
export TMPDIR=/tmp
O=gpurun_out/r04_52; mkdir -p $O
( timeout 1500 python scripts/fuzz_near_adversarial.py --n 2600 --seed 46001 --oracle-every 2 --out $O/fuzz_near_46001.jsonl 2> $O/fuzz_near_46001.err ); tail -1 $O/fuzz_near_46001.jsonl | cut -c1-400
( timeout 1500 python scripts/fuzz_near_adversarial.py --n 2600 --seed 46002 --oracle-every 2 --out $O/fuzz_near_46002.jsonl 2> $O/fuzz_near_46002.err ); tail -1 $O/fuzz_near_46002.jsonl | cut -c1-400
( HZ_FUZZ_N=300 HZ_FUZZ_SEED=46003 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -x -k "not stray" > $O/fuzz_46003.log 2>&1 ); tail -1 $O/fuzz_46003.log
