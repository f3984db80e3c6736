#!/bin/bash
# Builds vcr_gaus_amd/libvcr_raster_<tag>.so from the default objects (make -C vcr_gaus_amd/csrc first) with ONE translation unit
# recompiled under extra flags:   bash profiles/r6_build_variant.sh <tag> <unit.hip> "<extra flags>"
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
tag=$1; unit=$2; extra=$3
B=$R/build/var_$tag
mkdir -p $B
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-math-errno -fno-slp-vectorize -Wall -Wno-unused-function \
    $extra -c $R/vcr_gaus_amd/csrc/$unit -o $B/${unit%.hip}.o
objs=""
for u in capi preprocess binning composite radix_sort losses model_ops; do
    if [ "$u.hip" == "$unit" ]; then objs="$objs $B/$u.o"; else objs="$objs $R/build/csrc/$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/vcr_gaus_amd/libvcr_raster_$tag.so $objs
echo built libvcr_raster_$tag.so
