#!/bin/bash
# round 5, GPU call 16: wide sweeps on the final code: adversarial certificate configurations 2 x 2600 (every
# shortened ray re-traced, every second configuration against the oracle: horizon, ray and guard counts), 1000 random
# configurations (tests/test_gpu_fuzz.py with HZ_FUZZ_N=1000: against the oracle's tree), and the counter-example replay
export TMPDIR=/tmp
O=gpurun_out/r05_16; mkdir -p $O
( time timeout 600 python -c "import torch; print(torch.__version__)" ) > $O/torch_import.log 2>&1
( timeout 300 python scripts/replay_adv.py 48001 2536 > $O/replay_adv_48001_2536.log 2>&1 ); cat $O/replay_adv_48001_2536.log | cut -c1-250
( timeout 2400 python scripts/fuzz_near_adversarial.py --n 2600 --seed 52001 --oracle-every 2 --out $O/fuzz_near_52001.jsonl 2> $O/fuzz_near_52001.err ); tail -1 $O/fuzz_near_52001.jsonl | cut -c1-400
( timeout 2400 python scripts/fuzz_near_adversarial.py --n 2600 --seed 52002 --oracle-every 2 --out $O/fuzz_near_52002.jsonl 2> $O/fuzz_near_52002.err ); tail -1 $O/fuzz_near_52002.jsonl | cut -c1-400
( HZ_FUZZ_N=1000 HZ_FUZZ_SEED=52003 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -q -x -k "not stray" --durations=6 > $O/fuzz_52003.log 2>&1 ); tail -9 $O/fuzz_52003.log
