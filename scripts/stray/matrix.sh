#!/bin/bash
# Runs the stray-element replay (scripts/stray/replay.py) N times per variant and logs the outcomes.
# usage: scripts/stray/matrix.sh <outdir> [runs-per-variant]
out=${1:-gpurun_out/stray}; n=${2:-5}
mkdir -p $out
gcc -O1 -g -fPIC -shared -o scripts/stray/libhzq.so scripts/stray/hzq_preload.c -ldl -lpthread || exit 1
hzq=$PWD/scripts/stray/libhzq.so
run() {   # tag, env...
    tag=$1; shift
    fails=0
    for i in $(seq 1 $n); do
        env "$@" timeout 120 python scripts/stray/replay.py > $out/$tag.$i.log 2>&1
        rc=$?
        echo "$tag run $i rc=$rc $(grep -c DIRTY $out/$tag.$i.log) dirty" >> $out/summary.txt
        [ $rc -ne 0 ] && fails=$((fails+1))
        # a tripwire hit ends the variant early: the log holds the backtraces
        if [ $rc -eq 97 ] || grep -q "DAMAGED" $out/$tag.$i.log; then break; fi
    done
    echo "== $tag: $fails non-zero exits" >> $out/summary.txt
}
nproc >> $out/summary.txt
run base X=1
run page LD_PRELOAD=$hzq
run fill LD_PRELOAD=$hzq HZQ_MODE=fill
run page_nobt LD_PRELOAD=$hzq HZQ_BT=0
run page_wide LD_PRELOAD=$hzq HZQ_MIN=64 HZQ_MAX=16384 HZQ_CAP=60000
cat $out/summary.txt
