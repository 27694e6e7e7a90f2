#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box (usage: scripts/profile_bench.sh <prefix>):
# kernel trace + stats, then FETCH_SIZE, WRITE_SIZE and the SQ instruction counters in their own PMC passes
# (never combined with a trace domain).  Outputs under gpurun_out/<prefix>_*; scripts/refresh_profiles.py
# copies the summaries into profiles/.
P=$1
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${P}_kt -- \
    python $R/bench.py --no-cpu-baseline > $R/gpurun_out/${P}_kt_bench.json 2> $R/gpurun_out/${P}_kt.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${P}_fetch -- \
    python $R/bench.py --steps 2 --no-cpu-baseline --no-count --no-peaks > /dev/null 2> $R/gpurun_out/${P}_fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${P}_write -- \
    python $R/bench.py --steps 2 --no-cpu-baseline --no-count --no-peaks > /dev/null 2> $R/gpurun_out/${P}_write.err
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/${P}_sq -- \
    python $R/bench.py --steps 2 --no-cpu-baseline --no-peaks > $R/gpurun_out/${P}_sq_bench.json 2> $R/gpurun_out/${P}_sq.err
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${P}_grbm -- \
    python $R/bench.py --steps 2 --no-cpu-baseline --no-count --no-peaks > /dev/null 2> $R/gpurun_out/${P}_grbm.err
cd $R
tail -1 gpurun_out/${P}_kt_bench.json | cut -c1-400
