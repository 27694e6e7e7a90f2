#!/bin/bash
# After the stream-pool fix: N plain replays, N under the page tripwire, N under the fill tripwire.
out=${1:-gpurun_out/stray_fix}; n=${2:-6}
mkdir -p $out
gcc -O1 -g -fPIC -shared -o scripts/stray/libhzq.so scripts/stray/hzq_preload.c -ldl -lpthread || exit 1
hzq=$PWD/scripts/stray/libhzq.so
for tag in base page fill; do
  for i in $(seq 1 $n); do
    case $tag in
      base) env X=1 timeout 150 python scripts/stray/replay.py > $out/$tag.$i.log 2>&1;;
      page) env LD_PRELOAD=$hzq HZQ_CAP=8000 timeout 150 python scripts/stray/replay.py > $out/$tag.$i.log 2>&1;;
      fill) env LD_PRELOAD=$hzq HZQ_MODE=fill HZQ_CAP=8000 timeout 150 python scripts/stray/replay.py > $out/$tag.$i.log 2>&1;;
    esac
    echo "$tag run $i rc=$? dirty=$(grep -c DIRTY $out/$tag.$i.log) damaged=$(grep -c DAMAGED $out/$tag.$i.log) uaf=$(grep -c 'use after free' $out/$tag.$i.log)" >> $out/summary.txt
  done
done
cat $out/summary.txt
