#!/bin/bash
# SQ counters of k_shadow over the config-4 probe (scripts/bench_shadow.py --suns 24): lane utilisation, VALU busy
# share.  usage: scripts/pmc_shadow.sh <outdir>   (separate PMC passes, no trace domains)
out=$1
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/$out
cd /tmp
for set in "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -c1-12 | tr " " "_")
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $R/$out/$n -- python $R/scripts/bench_shadow.py --suns 24 > /dev/null 2> $R/$out/$n.err
done
cd $R
python - <<PY
import csv, glob, json
res = {}
for f in glob.glob("$out/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_shadow" in r["Kernel_Name"]:
            res.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in res.items()}
m["dispatches"] = len(next(iter(res.values()))) if res else 0
if "SQ_INSTS_VALU" in m:
    m["lane_utilisation_valu"] = m["SQ_THREAD_CYCLES_VALU"] / 64.0 / m["SQ_INSTS_VALU"]
    if "GRBM_GUI_ACTIVE" in m:
        m["valu_issue_slots_used"] = 4.0 * m["SQ_INSTS_VALU"] / (1024 * m["GRBM_GUI_ACTIVE"] / 8.0)
json.dump(m, open("$out/summary.json", "w"), indent=1)
print(json.dumps(m))
PY
