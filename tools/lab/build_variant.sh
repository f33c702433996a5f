#!/usr/bin/env bash
# Lab: build a variant of the library with extra -D flags into lab_build/libe3d_<name>.so (only the listed
# sources are recompiled with the flags; the other objects are the product's).
#   tools/lab/build_variant.sh <name> "<flags>" <source.hip> [more sources]
set -euo pipefail
cd "$(dirname "$0")/../../e3-invaraint-diffusion-model_amd/csrc"
NAME=$1; FLAGS=$2; shift 2
mkdir -p ../../lab_build/obj
OBJS=""
for o in *.o; do
  keep=1
  for s in "$@"; do [ "${s%.hip}.o" = "$o" ] && keep=0; done
  [ $keep = 1 ] && OBJS="$OBJS $o"
done
for s in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function $FLAGS -c "$s" -o ../../lab_build/obj/${NAME}_${s%.hip}.o
  OBJS="$OBJS ../../lab_build/obj/${NAME}_${s%.hip}.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o ../../lab_build/libe3d_${NAME}.so
echo "built lab_build/libe3d_${NAME}.so"
