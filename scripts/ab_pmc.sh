#!/bin/bash
# A/B of two builds of the library on one box: SQ counters of the horizon kernel over 2 bench steps
# usage: scripts/ab_pmc.sh <outdir> <dirA> <dirB>   (each dir holds bench.py + a built horayzon_amd/)
out=$1; shift
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/$out
cd /tmp
for d in "$@"; do
  tag=$(basename $d)
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_BRANCH" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
    n=$(echo $set | cut -c1-12 | tr " " "_")
    timeout 600 rocprofv3 --pmc $set --output-format csv -d $R/$out/${tag}_$n -- python $R/$d/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-count --no-peaks > /dev/null 2> $R/$out/${tag}_$n.err
  done
done
cd $R
python - <<PY
import csv, glob, json
res = {}
for f in glob.glob("$out/*/*/*counter_collection.csv"):
    tag = f.split("/")[-3].split("_SQ")[0].split("_TCC")[0]
    for r in csv.DictReader(open(f)):
        if "k_horizon<2, false, true, false" in r["Kernel_Name"]:
            res.setdefault(tag, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
out = {t: {k: sum(v) / len(v) for k, v in d.items()} for t, d in res.items()}
json.dump(out, open("$out/summary.json", "w"), indent=1)
keys = sorted({k for d in out.values() for k in d})
print("%-30s" % "counter", *["%14s" % t for t in out])
for k in keys: print("%-30s" % k, *["%14.5g" % out[t].get(k, float("nan")) for t in out])
PY
