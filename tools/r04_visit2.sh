#!/bin/bash
# GPU visit 2 of round 4: new kernels (fp16 indirect path) -- parity tests + precision probe
set -u
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"; cd "$REPO"
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > "$OUT/v2_parity.log" 2>&1; echo "parity rc=$?"; tail -25 "$OUT/v2_parity.log"
timeout -k 5 900 python tools/prec_probe.py "$OUT/v2_prec_probe.json" > "$OUT/v2_prec_probe.log" 2>&1; echo "prec rc=$?"; tail -40 "$OUT/v2_prec_probe.log"
