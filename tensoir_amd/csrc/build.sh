#!/bin/bash
# Build libtensoir_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [outdir]
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${1:-$HERE/..}"
INC="$HERE/../../include"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$INC -I$HERE -Wall -Wno-unused-function"
mkdir -p "$HERE/obj"
pids=()
for f in tir_field tir_march tir_mlp tir_shade tir_train; do
  if [ ! -f "$HERE/obj/$f.o" ] || [ "$HERE/$f.hip" -nt "$HERE/obj/$f.o" ] || [ "$HERE/tir_common.hpp" -nt "$HERE/obj/$f.o" ] || [ "$INC/tensoir_hip.h" -nt "$HERE/obj/$f.o" ]; then
    $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$HERE/obj/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC "$HERE"/obj/tir_field.o "$HERE"/obj/tir_march.o "$HERE"/obj/tir_mlp.o "$HERE"/obj/tir_shade.o "$HERE"/obj/tir_train.o -o "$OUT/libtensoir_hip.so"
echo "built $OUT/libtensoir_hip.so"
