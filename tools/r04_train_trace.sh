#!/bin/bash
# training step: bench line + kernel trace -> per-step GPU busy / idle (tools/trace_step.py)
set -u
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"; cd "$REPO"
timeout -k 5 600 python bench.py --workload train --steps 100 --warmup 5 --no-cpu-baseline > "$OUT/v5_train.json" 2> "$OUT/v5_train.err"; echo "train rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/v5_train.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','hip_ms_per_step')}, d['roofline']['frac'], d['roofline'].get('combining_factor'))
for e in d['entry_points']: print(e)
PY
P="$OUT/prof_v5_train"; mkdir -p "$P"; cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$P" -o trace -- python $REPO/bench.py --workload train --steps 30 --warmup 2 --no-cpu-baseline > "$P/trace.log" 2>&1
cd "$REPO"; python tools/trace_step.py "$P" 2>&1 | tail -40
find "$P" -name "*.db" -delete; find "$P" -name "*kernel_trace.csv" -size +6M -delete
