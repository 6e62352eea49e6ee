#!/bin/bash
# An A/B build of the WHOLE library with other compiler flags (e.g. -ffp-contract=fast): every translation unit recompiled.
#   bash scripts/build_all_variant.sh <tag> <flags ...>   ->  dpilqr_amd/variants/libdpilqr_hip_<tag>.so   (select with DPILQR_DEBUG_ROUTES=1 DPILQR_LIB=...)
# The given flags come after the library's own, so a repeated option (-ffp-contract=...) overrides it.
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/dpilqr_amd/variants $root/build/variants
flags=$(cd $root && python -c "import __graft_entry__ as g; print(' '.join(g.HIPCC_FLAGS))" | tail -1)
cd $root   # the flags name include paths relative to the repository
objs=""; pids=""
for src in $root/dpilqr_amd/csrc/*.hip; do
  u=$(basename $src .hip); o=$root/build/variants/${u}_$tag.o
  uflags=$(python -c "import __graft_entry__ as g; print(' '.join(g.UNIT_FLAGS.get('$u', [])))" | tail -1)
  /opt/rocm/bin/hipcc $flags $uflags "$@" -c -o $o $src &
  pids="$pids $!"; objs="$objs $o"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/dpilqr_amd/variants/libdpilqr_hip_$tag.so $objs
echo $root/dpilqr_amd/variants/libdpilqr_hip_$tag.so
