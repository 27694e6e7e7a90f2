#!/bin/bash
# Build a VARIANT of the library next to the product one, for same-box A/B runs:
#   scripts/build_variant.sh <name> "<extra hipcc flags, e.g. -DHZ_PROBE_EXTRA_LOADS=2>"
# -> horayzon_amd/libhorayzon_hip_<name>.so (git-ignored; travels with the gpurun snapshot).  Select it at run time with
#   HORAYZON_HIP_LIB=horayzon_amd/libhorayzon_hip_<name>.so python scripts/quick_perf.py ...
set -e
name=$1; shift
flags="$*"
R=$(cd "$(dirname "$0")/.." && pwd)
obj=/tmp/hz_variant_$name
mkdir -p $obj
cd $R/horayzon_amd/csrc
for f in hz_api hz_scene hz_horizon hz_shadow hz_locations hz_prep hz_sort hz_bench hz_near; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-result $flags -c $f.hip -o $obj/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/horayzon_amd/libhorayzon_hip_$name.so $obj/*.o
echo built horayzon_amd/libhorayzon_hip_$name.so
