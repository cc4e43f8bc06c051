#!/bin/bash
# A/B of env-var switches on the same box: tools/ab.sh VAR v1 v2 ...   (GPU box)
VAR=$1; shift
for v in "$@" "$1"; do
  env $VAR=$v python bench.py --no-exact-pass --no-cpu-baseline --no-graph 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$VAR=$v', d['value'], d['ms_per_step'], d['gpu_kernel_ms_per_step'], ' '.join(f\"{k['kernel'].replace('tir_','')}={k['ms_per_step']:.3f}\" for k in d['kernels'][:4]))
"
done
