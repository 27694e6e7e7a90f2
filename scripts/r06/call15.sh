#!/bin/bash
# round 6, GPU call 15: last check of the final tree: smoke(), the bench line (no extras), the config-4 line
O=gpurun_out/r06_15
mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py --no-extras > $O/bench_no_extras.json 2> $O/bench.err; tail -c 300 $O/bench_no_extras.json
timeout 600 python bench.py --workload c4 > $O/bench_c4.json 2>> $O/bench.err; tail -c 200 $O/bench_c4.json
