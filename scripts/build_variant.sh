#!/bin/bash
# An A/B build of the library: one translation unit recompiled with extra flags, linked with the current objects of the others.
#   bash scripts/build_variant.sh <tag> <unit (e.g. tu_forward)> <flags ...>   ->  dpilqr_amd/variants/libdpilqr_hip_<tag>.so
# Select it at run time with DPILQR_DEBUG_ROUTES=1 DPILQR_LIB=$PWD/dpilqr_amd/variants/libdpilqr_hip_<tag>.so (dpilqr_amd/_lib.py).  The directory is
# git-ignored but travels to the GPU box.
set -e
tag=$1; unit=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/dpilqr_amd/variants $root/build/variants
# every translation unit's object must exist and be current (build() alone compiles nothing when the linked library is newer
# than the sources, and objects do not travel to the GPU box)
flags=$(cd $root && python -c "import __graft_entry__ as g; g.build(need_objects=True); print(' '.join(g.HIPCC_FLAGS + g.UNIT_FLAGS.get('$unit', [])))" | tail -1)
cd $root   # the flags name include paths relative to the repository
/opt/rocm/bin/hipcc $flags "$@" -c -o $root/build/variants/${unit}_$tag.o $root/dpilqr_amd/csrc/$unit.hip
objs=""
for src in $root/dpilqr_amd/csrc/*.hip; do
  u=$(basename $src .hip); o=$root/build/obj/$u.o
  if [ "$u" == "$unit" ]; then objs="$objs $root/build/variants/${unit}_$tag.o"; continue; fi
  if [ ! -f $o ] || [ $o -ot $src ]; then echo "build_variant.sh: object $o is missing or older than its source" >&2; exit 1; fi
  objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/dpilqr_amd/variants/libdpilqr_hip_$tag.so $objs
echo $root/dpilqr_amd/variants/libdpilqr_hip_$tag.so
