#!/bin/bash
# GPU visit: fused indirect kernel -- correctness tests, then bench with fused on / off
set -u
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"; cd "$REPO"
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "precision_policy or graph or renderer_boundary or c4 or importance" > "$OUT/v6_parity.log" 2>&1; echo "parity rc=$?"; tail -12 "$OUT/v6_parity.log"
cat > /tmp/pb.py <<'PY'
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], {k:d[k] for k in ('value','ms_per_step')}, 'single', d['single_stream']['value'], 'proto', d['protocol_2_1']['rays_per_s'], 'parity', d['parity']['max_abs'] if d.get('parity') else None, d['precision_policy'].get('fused_gather_decoder'))
for k in d['kernels'][:6]: print('   ', k['kernel'], round(k['avg_ms'],4), round(k.get('frac',0),4), k.get('decoder_frac_of_dense_fp16_peak'))
PY
for fu in 1 0; do
  TENSOIR_FUSED_INDIRECT=$fu timeout -k 5 600 python bench.py --no-side-workloads --no-sharp-scene --no-exact-pass --boundary-calls 20 --cpu-rays 512 --cpu-calls 1 2> "$OUT/v6_bench_$fu.err" | python /tmp/pb.py fused=$fu
done
