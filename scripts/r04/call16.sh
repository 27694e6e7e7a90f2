#!/bin/bash
# round 4, GPU call 16: wide random sweeps on the final code (new node layout / quantisation): gridded, locations, shadow
export TMPDIR=/tmp
O=gpurun_out/r04_16; mkdir -p $O
for seed in 43001 43002; do
  ( HZ_FUZZ_N=1500 HZ_FUZZ_SEED=$seed timeout 2400 python -m pytest tests/test_gpu_fuzz.py -q -x -k "not stray" > $O/fuzz_$seed.log 2>&1 ); echo "seed $seed: $(tail -1 $O/fuzz_$seed.log)"
done
