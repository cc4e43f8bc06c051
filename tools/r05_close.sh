#!/bin/bash
# Closing visit of round 5 for a tree whose LIBRARY is unchanged (same source hash as the committed traces / PMC files) but whose
# host side changed: the -m gpu suite (launcher tests too when a checkout is staged), the default bench line as the driver runs
# it, the training line + its kernel trace, the unmodified script's iteration times.  Usage (via gpurun): tools/r05_close.sh <tag>
set -u
TAG="${1:-r05close}"
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$REPO/gpurun_out"; mkdir -p "$OUT"; cd "$REPO"
[ -f "$REPO/gpurun_scratch/reference/train_tensoIR.py" ] && export TENSOIR_REFERENCE="$REPO/gpurun_scratch/reference"
timeout -k 5 900 python -m pytest tests -m gpu -q > "$OUT/${TAG}_tests.log" 2>&1
echo "tests rc=$?" >> "$OUT/${TAG}_tests.log"; tail -4 "$OUT/${TAG}_tests.log"
env -u TENSOIR_REFERENCE timeout -k 5 600 python bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
echo "bench rc=$?"; tail -c 300 "$OUT/${TAG}_bench.json"; echo
timeout -k 5 300 python bench.py --workload train > "$OUT/${TAG}_train_bench.json" 2> "$OUT/${TAG}_train_bench.err"; echo "train rc=$?"
python -c "import json; d=json.loads([l for l in open('$OUT/${TAG}_train_bench.json') if l.startswith('{')][-1]); print('train', d['ms_per_step'], d['parity']['ok'])"
if [ -n "${TENSOIR_REFERENCE:-}" ]; then
  timeout -k 5 400 python tools/script_head_to_head.py --out "$OUT/${TAG}_script_hip.json" --modes hip > /dev/null 2>&1
  python -c "import json; d=json.load(open('$OUT/${TAG}_script_hip.json')); print('script', d['hip']['ms_per_iteration'])"
fi
P="$OUT/prof_${TAG}"; mkdir -p "$P"; cd /tmp; export TMPDIR=/tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/trace_train" -o trace -- python $REPO/bench.py --workload train --steps 3 --warmup 1 --no-cpu-baseline > "$P/trace_train.log" 2>&1
echo "trace train rc=$?"
f=$(find "$P/trace_train" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${TAG}_kernel_stats_train.csv"
find "$P" -name "*.db" -delete; find "$P" -name "*kernel_trace.csv" -delete; du -sh "$OUT"
