#!/bin/bash
# round 6, GPU call 9: replay of the failing adversarial sweep (seed 64003, two hand-over levels), then the two-level A/B
O=gpurun_out/r06_09
mkdir -p $O
timeout 2400 python scripts/r06/replay_64003.py 2600 --all > $O/replay_64003.log 2>&1
tail -5 $O/replay_64003.log | cut -c1-1500
