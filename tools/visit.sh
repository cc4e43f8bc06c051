#!/bin/bash
# One GPU-box visit (parametrised).  Usage (via gpurun):
#   tools/visit.sh <tag> <step> [<step> ...]     steps: tests | tests:<pytest args> | precision | bench | bench:<args> | cmd:<shell command>
# Everything lands in gpurun_out/<tag>_*.
set -u
TAG="$1"; shift
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$REPO/gpurun_out"; mkdir -p "$OUT"; cd "$REPO"
export TMPDIR=/tmp
for step in "$@"; do
  name="${step%%:*}"; arg=""; [ "$name" != "$step" ] && arg="${step#*:}"
  t0=$(date +%s)
  case "$name" in
    tests) # arg = pytest arguments (files, -k 'a or b' with its own quotes); none = the whole -m gpu suite
           if [ -z "$arg" ]; then arg="tests"; fi
           eval "timeout -k 5 1500 python -m pytest -m gpu -q $arg" > "$OUT/${TAG}_tests.log" 2>&1; echo "tests rc=$?" >> "$OUT/${TAG}_tests.log"; tail -4 "$OUT/${TAG}_tests.log";;
    precision) timeout -k 5 1200 python tools/precision_sweep.py ${arg:-} > "$OUT/${TAG}_precision.log" 2>&1; echo "precision rc=$?"; tail -25 "$OUT/${TAG}_precision.log";;
    bench) timeout -k 5 900 python bench.py ${arg:-} > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"; echo "bench rc=$?"; tail -c 700 "$OUT/${TAG}_bench.json"; echo; tail -3 "$OUT/${TAG}_bench.err";;
    cmd) bash -c "$arg" > "$OUT/${TAG}_cmd.log" 2>&1; echo "cmd rc=$?"; tail -30 "$OUT/${TAG}_cmd.log";;
    *) echo "unknown step $step";;
  esac
  echo "[$name: $(( $(date +%s) - t0 )) s]"
done
