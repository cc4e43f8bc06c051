#!/bin/bash
# tools/build_variant2.sh <name> <-D flags...>: rebuilds tir_field + tir_march with the flags -> gpurun_scratch/lib_<name>.so
set -e
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
NAME=$1; shift
C=$REPO/tensoir_amd/csrc
mkdir -p $REPO/gpurun_scratch
for f in tir_field tir_march; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$REPO/include -I$C -Wno-unused-function "$@" -c $C/$f.hip -o /tmp/${f}_$NAME.o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/tir_field_$NAME.o /tmp/tir_march_$NAME.o $C/obj/tir_mlp.o $C/obj/tir_shade.o $C/obj/tir_train.o -o $REPO/gpurun_scratch/lib_$NAME.so
echo built lib_$NAME.so
