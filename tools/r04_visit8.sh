#!/bin/bash
set -u
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"; cd "$REPO"
timeout -k 5 1500 python -m pytest tests -m gpu -q -x > "$OUT/v8_tests.log" 2>&1; echo "tests rc=$?"; tail -6 "$OUT/v8_tests.log"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for wl in image relight train; do
  timeout -k 5 500 python bench.py --workload $wl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', d['value'], d['ms_per_step'], d['parity']['ok'], d['roofline']['kernel'], d['roofline']['frac'])"
done
