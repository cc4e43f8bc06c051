#!/bin/bash
# Re-run the tests and the four bench lines only (after the PMC files of the library being timed are in place: `pmc.stale` false).
# Usage (via gpurun): tools/r05_rebench.sh <tag>
set -u
TAG="${1:-r05g}"
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"; cd "$REPO"
unset TENSOIR_REFERENCE
timeout -k 5 1500 python -m pytest tests -m gpu -q > "$OUT/${TAG}_tests.log" 2>&1; echo "tests rc=$?" >> "$OUT/${TAG}_tests.log"; tail -4 "$OUT/${TAG}_tests.log"
cp "$OUT/parity_fullsize.json" "$OUT/${TAG}_parity_fullsize.json" 2>/dev/null
timeout -k 5 900 python bench.py --breakdown "$OUT/${TAG}_breakdown.json" > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
echo "bench rc=$?"; tail -c 300 "$OUT/${TAG}_bench.json"; echo; tail -3 "$OUT/${TAG}_bench.err"
timeout -k 5 500 python bench.py --workload image --simulate-ranks 8 > "$OUT/${TAG}_image_bench.json" 2> "$OUT/${TAG}_image_bench.err"; echo "image rc=$?"
timeout -k 5 500 python bench.py --workload relight --simulate-ranks 8 > "$OUT/${TAG}_relight_bench.json" 2> "$OUT/${TAG}_relight_bench.err"; echo "relight rc=$?"
timeout -k 5 300 python bench.py --workload relight --steps 5 --c5-host-masking --no-cpu-baseline > "$OUT/${TAG}_relight_host_masking_bench.json" 2> "$OUT/${TAG}_relight_host_masking_bench.err"; echo "relight host rc=$?"
timeout -k 5 500 python bench.py --workload train --steps 100 --warmup 5 > "$OUT/${TAG}_train_bench.json" 2> "$OUT/${TAG}_train_bench.err"; echo "train rc=$?"
for wl in image relight train; do tail -c 200 "$OUT/${TAG}_${wl}_bench.json"; echo; done
if [ -f "$REPO/gpurun_scratch/reference/train_tensoIR.py" ]; then
  export TENSOIR_REFERENCE="$REPO/gpurun_scratch/reference"
  timeout -k 5 400 python tools/script_head_to_head.py --out "$OUT/${TAG}_script_hip.json" --modes hip > /dev/null 2>&1
  python -c "import json; d=json.load(open('$OUT/${TAG}_script_hip.json')); print('script', d['hip']['ms_per_iteration'], d.get('render_test', {}).get('hip', {}).get('s_per_image'))"
fi
