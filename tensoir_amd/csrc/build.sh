#!/bin/bash
# Build libtensoir_hip.so for gfx950 (cross-compiles without a GPU).
# Usage: build.sh [--force] [outdir]
#
# An object is reused only when the sha256 of everything that goes into it (its .hip source, tir_common.hpp, the ABI
# header, the compiler flags, the hipcc version string) equals the stamp written next to it by the compile that
# produced it -- objects shipped in a snapshot (git-ignored, but they travel with gpurun) never satisfy a build by their
# mtime alone.  --force (or TENSOIR_FORCE_BUILD=1) recompiles every source regardless.  The line "compiled: ... reused: ..."
# says what this invocation did; csrc/obj/BUILD_STAMP records the source hash the library was linked from
# (tir_source_hash(), bench.py: `library.source_hash`).
set -euo pipefail
FORCE="${TENSOIR_FORCE_BUILD:-0}"
if [ "${1:-}" = "--force" ]; then FORCE=1; shift; fi
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${1:-$HERE/..}"
INC="$HERE/../../include"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$INC -I$HERE -Wall -Wno-unused-function ${TENSOIR_EXTRA_FLAGS:-}"
SRCS="tir_field tir_march tir_mlp tir_shade tir_train"
mkdir -p "$HERE/obj"
HIPVER="$($HIPCC --version 2>/dev/null | head -3 | sha256sum | cut -c1-16)"
# the flags enter the stamp without the absolute checkout path (a snapshot on another box must not look like other flags)
FLAGKEY="$(echo "$FLAGS" | sed "s#$HERE#CSRC#g; s#$INC#INC#g")"
src_hash() {   # $1 = source stem
  cat "$HERE/$1.hip" "$HERE/tir_common.hpp" "$INC/tensoir_hip.h" <(echo "$FLAGKEY $HIPVER") | sha256sum | cut -c1-32
}
pids=(); compiled=(); reused=()
for f in $SRCS; do
  want="$(src_hash "$f")"
  have="$(cat "$HERE/obj/$f.sha" 2>/dev/null || true)"
  if [ "$FORCE" = "1" ] || [ ! -f "$HERE/obj/$f.o" ] || [ "$want" != "$have" ]; then
    rm -f "$HERE/obj/$f.sha"
    ( $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$HERE/obj/$f.o" && echo "$want" > "$HERE/obj/$f.sha" ) &
    pids+=($!); compiled+=("$f")
  else
    reused+=("$f")
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
objs=""; for f in $SRCS; do objs="$objs $HERE/obj/$f.o"; done
# whole-library source hash: exported to the bench / the PMC stamps through csrc/obj/BUILD_STAMP and the .so's sidecar file
ALL="$(for f in $SRCS; do cat "$HERE/obj/$f.sha"; done | sha256sum | cut -c1-16)"
$HIPCC --offload-arch=gfx950 -shared -fPIC $objs -o "$OUT/libtensoir_hip.so"
echo "$ALL" > "$HERE/obj/BUILD_STAMP"
echo "$ALL" > "$OUT/libtensoir_hip.so.srchash"
echo "compiled: ${compiled[*]:-none}  reused: ${reused[*]:-none}  source_hash: $ALL"
echo "built $OUT/libtensoir_hip.so"
