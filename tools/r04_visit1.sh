#!/bin/bash
# GPU visit 1 of round 4: multirank tests (shared-GPU gloo), decoder parity tests, precision probe, self-launched 2-rank bench
set -u
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"; cd "$REPO"
timeout -k 5 600 python -m pytest tests/test_gpu_multirank.py -m gpu -q -x > "$OUT/v1_multirank.log" 2>&1; echo "multirank rc=$?"; tail -15 "$OUT/v1_multirank.log"
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "decoder or mlp or golden or graph" > "$OUT/v1_parity.log" 2>&1; echo "parity rc=$?"; tail -5 "$OUT/v1_parity.log"
timeout -k 5 900 python tools/prec_probe.py "$OUT/v1_prec_probe.json" > "$OUT/v1_prec_probe.log" 2>&1; echo "prec rc=$?"; tail -60 "$OUT/v1_prec_probe.log"
env -u WORLD_SIZE -u RANK -u LOCAL_RANK timeout 500 python bench.py --gpus 2 --backend gloo --allow-shared-gpu --steps 4 --warmup 1 --no-exact-pass --no-sharp-scene --no-cpu-baseline --sustained-steps 8 > "$OUT/v1_two_rank.out" 2> "$OUT/v1_two_rank.err"; echo "two-rank rc=$?"
tail -c 600 "$OUT/v1_two_rank.out"; grep "\[bench\]" "$OUT/v1_two_rank.err" | head
