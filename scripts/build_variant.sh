#!/bin/bash
# An A/B build of the library: one translation unit recompiled with extra flags, linked with the current objects of the others.
#   bash scripts/build_variant.sh <tag> <unit (e.g. tu_forward)> <flags ...>   ->  dpilqr_amd/variants/libdpilqr_hip_<tag>.so
# Select it at run time with DPILQR_LIB=$PWD/dpilqr_amd/variants/libdpilqr_hip_<tag>.so (dpilqr_amd/_lib.py).  The directory is
# git-ignored but travels to the GPU box.
set -e
tag=$1; unit=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/dpilqr_amd/variants $root/build/variants
python $root/__graft_entry__.py > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$root/include -I$root/dpilqr_amd/csrc "$@" \
    -c -o $root/build/variants/${unit}_$tag.o $root/dpilqr_amd/csrc/$unit.hip
objs=""
for o in $root/build/obj/*.o; do
  if [ "$(basename $o .o)" == "$unit" ]; then objs="$objs $root/build/variants/${unit}_$tag.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/dpilqr_amd/variants/libdpilqr_hip_$tag.so $objs
echo $root/dpilqr_amd/variants/libdpilqr_hip_$tag.so
