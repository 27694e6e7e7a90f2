#!/bin/bash
# Runs the stray-element replay (scripts/stray/replay.py) N times per variant and logs the outcomes.
# usage: scripts/stray/matrix.sh <outdir> [runs-per-variant]
out=${1:-gpurun_out/stray}; n=${2:-5}
mkdir -p $out
gcc -O1 -g -fPIC -shared -o scripts/stray/libhzq.so scripts/stray/hzq_preload.c -ldl -lpthread || exit 1
gcc -O1 -g -fPIC -shared -o scripts/stray/libhzwatch.so scripts/stray/hzwatch.c || exit 1
hzq=$PWD/scripts/stray/libhzq.so
run() {   # tag, replay args (quoted), env...
    tag=$1; rargs=$2; shift; shift
    fails=0
    for i in $(seq 1 $n); do
        env "$@" timeout 150 python scripts/stray/replay.py $rargs > $out/$tag.$i.log 2>&1
        rc=$?
        echo "$tag run $i rc=$rc dirty=$(grep -c DIRTY $out/$tag.$i.log) damaged=$(grep -c DAMAGED $out/$tag.$i.log) trapped=$(grep -c 'WRITE to watched' $out/$tag.$i.log)" >> $out/summary.txt
        [ $rc -ne 0 ] && fails=$((fails+1))
        # a tripwire hit ends the variant early: the log holds the backtraces
        if [ $rc -eq 97 ] || grep -q "DAMAGED" $out/$tag.$i.log; then break; fi
    done
    echo "== $tag: $fails non-zero exits" >> $out/summary.txt
}
nproc >> $out/summary.txt
run base "" X=1
run page "" LD_PRELOAD=$hzq HZQ_CAP=8000
run fill "" LD_PRELOAD=$hzq HZQ_MODE=fill HZQ_CAP=8000
run page_wide "" LD_PRELOAD=$hzq HZQ_MIN=48 HZQ_MAX=70000 HZQ_CAP=12000
run fill_wide "" LD_PRELOAD=$hzq HZQ_MODE=fill HZQ_MIN=48 HZQ_MAX=70000 HZQ_CAP=12000
run watch_post "--watch post" X=1
cat $out/summary.txt
