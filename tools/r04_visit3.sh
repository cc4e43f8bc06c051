#!/bin/bash
# GPU visit 3 of round 4: the whole -m gpu suite + the new default bench line (side workloads, full-batch CPU baseline)
set -u
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"; cd "$REPO"
( time timeout -k 5 900 python bench.py --breakdown "$OUT/v3_breakdown.json" > "$OUT/v3_bench.json" 2> "$OUT/v3_bench.err" ) 2> "$OUT/v3_bench.time"; echo "bench rc=$?"; tail -3 "$OUT/v3_bench.time"
tail -c 2500 "$OUT/v3_bench.json"; echo; tail -5 "$OUT/v3_bench.err"
timeout -k 5 1500 python -m pytest tests -m gpu -q -x > "$OUT/v3_tests.log" 2>&1; echo "tests rc=$?"; tail -15 "$OUT/v3_tests.log"
