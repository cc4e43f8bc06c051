#!/bin/bash
# A/B of launcher settings on the unmodified train_tensoIR.py (armadillo's hyper-parameters, compressed schedule), one GPU box:
# indirect-light policy x dataset residency, ms per iteration of the three phases.  Needs TENSOIR_REFERENCE (tools/stage_reference.sh).
# Usage (GPU box): tools/r05_script_ab.sh <tag> [policy:dataset ...]
TAG="${1:-ab}"; shift || true
[ $# -eq 0 ] && set -- auto:auto f16:auto auto:auto f16:auto
export TENSOIR_REFERENCE="${TENSOIR_REFERENCE:-$PWD/gpurun_scratch/reference}"
i=0
for v in "$@"; do
  p=${v%%:*}; dd=${v##*:}; i=$((i+1))
  TENSOIR_INDIRECT_PRECISION=$p TENSOIR_DEVICE_DATASET=$dd python tools/script_head_to_head.py --out gpurun_out/${TAG}_script_${i}_${p}_${dd}.json --modes hip --render 0 > /dev/null 2>&1
  python -c "import json; d=json.load(open('gpurun_out/${TAG}_script_${i}_${p}_${dd}.json')); print('$v', d['hip']['ms_per_iteration'], d['hip']['last_progress'])"
done
