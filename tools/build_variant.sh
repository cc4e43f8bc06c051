#!/bin/bash
# tools/build_variant.sh <file-stem> <name> <-D flags...>: builds gpurun_scratch/lib_<name>.so with one object recompiled
set -e
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
STEM=$1; NAME=$2; shift 2
C=$REPO/tensoir_amd/csrc
mkdir -p $REPO/gpurun_scratch
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$REPO/include -I$C -Wno-unused-function "$@" -c $C/$STEM.hip -o /tmp/${STEM}_$NAME.o 2>/dev/null
OBJS=""
for f in tir_field tir_march tir_mlp tir_shade tir_train; do
  if [ $f = $STEM ]; then OBJS="$OBJS /tmp/${STEM}_$NAME.o"; else OBJS="$OBJS $C/obj/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $REPO/gpurun_scratch/lib_$NAME.so
echo built lib_$NAME.so
