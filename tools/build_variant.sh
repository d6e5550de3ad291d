#!/bin/bash
# Links a library variant for A/B timing on the GPU box (tools/ab_variants.sh):
#   tools/build_variant.sh <unit>=<replacement.hip> [...] <out.so>      e.g.  blend=/tmp/blend_x.hip var/x.so
# Every other translation unit is taken from the objects of the regular build (run make first).
set -eu
CS=$(dirname "$0")/../g4splat_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-function -I$CS ${EXTRA:-}"
out="${@: -1}"
objs=""
declare -A repl
for a in "${@:1:$#-1}"; do repl[${a%%=*}]=${a#*=}; done
for u in api preprocess binning blend knn maps loss densify; do
  if [ -n "${repl[$u]:-}" ]; then
    cp "${repl[$u]}" $CS/_variant_$u.hip
    /opt/rocm/bin/hipcc $FLAGS -c $CS/_variant_$u.hip -o /tmp/_variant_$u.o
    rm -f $CS/_variant_$u.hip
    objs="$objs /tmp/_variant_$u.o"
  else
    objs="$objs $CS/$u.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" $objs
echo "built $out"
