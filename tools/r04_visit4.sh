#!/bin/bash
# GPU visit 4: march VALU diet -- bit-identity + parity tests, then the step timing (one lane, eager attribution)
set -u
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"; cd "$REPO"
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_parity.py -m gpu -q -x > "$OUT/v4_parity.log" 2>&1; echo "parity rc=$?"; tail -4 "$OUT/v4_parity.log"
timeout -k 5 600 python bench.py --no-cpu-baseline --no-side-workloads --no-sharp-scene --no-exact-pass --boundary-calls 5 > "$OUT/v4_bench.json" 2> "$OUT/v4_bench.err"; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/v4_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['single_stream'], d['protocol_2_1']['rays_per_s'])
for k in d['kernels']: print(k['kernel'], round(k['avg_ms'],4), round(k.get('frac',0),4))
PY
