#!/bin/bash
# PMC passes for the horizon kernel on the 1024^2 window probe (run on the GPU box).
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/pmc_$1
shift
mkdir -p $OUT
cd /tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH" \
           "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE TA_BUSY_avr"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python $R/scripts/quick_perf.py --reps 1 "$@" > $OUT/p$i.log 2>&1
done
cd $R
python - <<PY
import csv, glob, json
agg = {}
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_horizon" in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            agg["VGPR"] = [float(r["VGPR_Count"])]; agg["LDS"] = [float(r["LDS_Block_Size"])]; agg["SGPR"] = [float(r["SGPR_Count"])]
s = {k: sum(v) / len(v) for k, v in agg.items()}
json.dump(s, open("$OUT/summary.json", "w"), indent=1)
for k in sorted(s): print("%-32s %.4g" % (k, s[k]))
PY
grep "rep 0" $OUT/p1.log
