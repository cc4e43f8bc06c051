#!/bin/bash
# One GPU-box visit: -m gpu tests, the default bench line, a rocprofv3 kernel trace of the bench command.
# Usage (via gpurun): tools/gpu_round.sh <tag> [notests]   -> gpurun_out/<tag>_*
set -u
TAG="${1:-r02}"
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$REPO/gpurun_out"
mkdir -p "$OUT"
cd "$REPO"
if [ "${2:-}" != "notests" ]; then
  timeout -k 5 1500 python -m pytest tests -m gpu -q > "$OUT/${TAG}_tests.log" 2>&1
  echo "tests rc=$?" >> "$OUT/${TAG}_tests.log"
  tail -5 "$OUT/${TAG}_tests.log"
  cp "$OUT/parity_fullsize.json" "$OUT/${TAG}_parity_fullsize.json" 2>/dev/null
fi
timeout -k 5 900 python bench.py --breakdown "$OUT/${TAG}_breakdown.json" > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
echo "bench rc=$?"; tail -c 1500 "$OUT/${TAG}_bench.json"; tail -5 "$OUT/${TAG}_bench.err"
mkdir -p "$OUT/prof_${TAG}"
cd /tmp && export TMPDIR=/tmp
# --in-flight 1: one batch at a time, so that a kernel's duration in the trace is its own (the roofline rows of the bench are
# measured the same way); the second trace is the default command (two batches in flight: durations overlap each other)
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${TAG}/trace" -o trace -- \
  python "$REPO/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 1 --boundary-calls 3 --no-exact-pass --no-sharp-scene --in-flight 1 > "$OUT/prof_${TAG}/trace.log" 2>&1
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${TAG}/trace2" -o trace -- \
  python "$REPO/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 1 --boundary-calls 3 --no-exact-pass --no-sharp-scene > "$OUT/prof_${TAG}/trace2.log" 2>&1
find "$OUT/prof_${TAG}" -name "*.db" -size +20M -delete
python "$REPO/tools/summarize_prof.py" "$OUT/prof_${TAG}/trace" > "$OUT/prof_${TAG}/summary.txt" 2>&1
python "$REPO/tools/summarize_prof.py" "$OUT/prof_${TAG}/trace2" > "$OUT/prof_${TAG}/summary_inflight2.txt" 2>&1
head -40 "$OUT/prof_${TAG}/summary.txt"
