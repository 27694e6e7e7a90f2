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
# kernel trace + stats of the timed launches (whole tile per launch).  --no-e2e: the untimed drop-in call at the end of
# the default command launches the same kernel on 5 chunks of rows, which would mix into the per-kernel average; the
# default command (with it and with the untimed extras except config 5: binary_search, discrete_sampling, the curved tile, c4)
# is traced as well, into ${P}_kte -- its per-kernel table is the rocprofv3 evidence for the ALG 0 / 1 instantiations
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${P}_kt -- \
    python $R/bench.py --no-cpu-baseline --no-e2e --no-extras > $R/gpurun_out/${P}_kt_bench.json 2> $R/gpurun_out/${P}_kt.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${P}_kte -- \
    python $R/bench.py --no-cpu-baseline --no-c5-extra > $R/gpurun_out/${P}_kte_bench.json 2> $R/gpurun_out/${P}_kte.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${P}_fetch -- \
    python $R/bench.py --steps 2 --no-cpu-baseline --no-count --no-peaks --no-e2e --no-extras > /dev/null 2> $R/gpurun_out/${P}_fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${P}_write -- \
    python $R/bench.py --steps 2 --no-cpu-baseline --no-count --no-peaks --no-e2e --no-extras > /dev/null 2> $R/gpurun_out/${P}_write.err
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/${P}_sq -- \
    python $R/bench.py --steps 2 --no-cpu-baseline --no-peaks --no-e2e --no-extras > $R/gpurun_out/${P}_sq_bench.json 2> $R/gpurun_out/${P}_sq.err
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${P}_grbm -- \
    python $R/bench.py --steps 2 --no-cpu-baseline --no-count --no-peaks --no-e2e --no-extras > /dev/null 2> $R/gpurun_out/${P}_grbm.err
# config 4 (shadow, 144 sun positions in one launch): kernel trace + stats, and the SQ counters of k_shadow_refill
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${P}_c4kt -- \
    python $R/bench.py --workload c4 > $R/gpurun_out/${P}_c4kt_bench.json 2> $R/gpurun_out/${P}_c4kt.err
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/${P}_c4sq -- \
    python $R/bench.py --workload c4 --steps 2 --no-count > /dev/null 2> $R/gpurun_out/${P}_c4sq.err
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${P}_c4grbm -- \
    python $R/bench.py --workload c4 --steps 2 --no-count > /dev/null 2> $R/gpurun_out/${P}_c4grbm.err
# wait / busy split of the wave cycles, final kernels (two passes each: the SQ counters of one pass share 8 slots)
for W in "wb1:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "wb2:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU"; do
  tag=${W%%:*}; set=${W#*:}
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/${P}_$tag -- \
      python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-count --no-peaks --no-e2e --no-extras > /dev/null 2> $R/gpurun_out/${P}_$tag.err
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/${P}_c4$tag -- \
      python $R/bench.py --workload c4 --steps 1 --warmup 0 --no-count > /dev/null 2> $R/gpurun_out/${P}_c4$tag.err
done
cd $R
tail -1 gpurun_out/${P}_kt_bench.json | cut -c1-400
