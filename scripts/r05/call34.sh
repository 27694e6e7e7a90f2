#!/bin/bash
# round 5, GPU call 34: the final tree (with the leftover cells) -- full GPU suite, the counter-example replay, random + adversarial sweeps with the block loop of the
# persistent kernel forced onto their small grids (HZ_PERSIST_GRID), rocprofv3 passes over bench.py (kernel trace, PMC), the default bench line
export TMPDIR=/tmp
O=gpurun_out/r05_34; mkdir -p $O
( time timeout 900 python -c "import torch; print(torch.__version__)" ) > $O/torch_import.log 2>&1
( timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > $O/tests_gpu_full.log 2>&1 ); tail -3 $O/tests_gpu_full.log
( timeout 300 python scripts/replay_adv.py 48001 2536 > $O/replay_adv_48001_2536.log 2>&1 ); cat $O/replay_adv_48001_2536.log | cut -c1-250
( HZ_PERSIST_GRID=5 HZ_FUZZ_N=300 HZ_FUZZ_SEED=55003 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -x -k "not stray" > $O/fuzz_55003_persist_grid5.log 2>&1 ); tail -2 $O/fuzz_55003_persist_grid5.log
( HZ_PERSIST_GRID=7 timeout 1500 python scripts/fuzz_near_adversarial.py --n 1300 --seed 55001 --oracle-every 2 --out $O/fuzz_near_55001.jsonl 2> $O/fuzz_near_55001.err ); tail -1 $O/fuzz_near_55001.jsonl | cut -c1-400
bash scripts/profile_bench.sh r05c
