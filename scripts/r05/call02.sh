#!/bin/bash
# round 5, GPU call 2: what does the box-test start cost?  same-box A/B of r4 / tau = 0 / 4 / 16 pads (whole C3 tile), with the
# work counters of each; then the SQ wait / busy counters VERDICT r4 item 4 asks for, and one more look for PC sampling
export TMPDIR=/tmp
R=$PWD
O=gpurun_out/r05_02; mkdir -p $O
for rep in 1 2; do
  for lib in r4 tau0 tau4 product; do
    if [ $lib = product ]; then unset HORAYZON_HIP_LIB; else export HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_$lib.so; fi
    ( timeout 300 python scripts/quick_perf.py --win 3569 --reps 3 --count > $O/perf_${lib}_$rep.log 2>&1 ); echo $lib $rep; grep "^rep\|SIMT" $O/perf_${lib}_$rep.log | cut -c1-200
  done
done
unset HORAYZON_HIP_LIB
cd /tmp
( timeout 120 rocprofv3 -L > $R/$O/avail.txt 2>&1 ); grep -c . $R/$O/avail.txt; grep -n -i "pc.sampl" $R/$O/avail.txt | head -5
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-60)
  ( timeout 400 rocprofv3 --pmc $set --output-format csv -d $R/$O/pmc_$tag -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-count --no-peaks --no-e2e --no-extras > /dev/null 2> $R/$O/pmc_$tag.err ); echo "pmc $tag exit $?"
done
cd $R
python - <<'PY'
import csv, glob, collections, json
out = {}
for f in glob.glob("gpurun_out/r05_02/pmc_*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:60]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    for k, v in acc.items():
        out.setdefault(k, {}).update(v)
json.dump(out, open("gpurun_out/r05_02/pmc_summary.json", "w"), indent=1)
for k, v in out.items():
    if "k_horizon" in k or "near" in k or "topo" in k: print(k, json.dumps(v))
PY
