#!/bin/bash
# One GPU-box visit with the staged reference checkout (tools/stage_reference.sh): the unmodified training / evaluation / relighting scripts end to
# end on the HIP path, then oracle/ref_on_gpu.py (reference on the host cores, reference on the MI355X as on-device oracle,
# mask maintenance, C5 at 400^3).  Usage (via gpurun): tools/ref_round.sh <tag> [extra ref_on_gpu.py flags]  -> gpurun_out/<tag>_*
set -u
TAG="${1:-r03}"; shift || true
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$REPO/gpurun_out"
mkdir -p "$OUT"
cd "$REPO"
export TENSOIR_REFERENCE="$REPO/gpurun_scratch/reference"
[ -f "$TENSOIR_REFERENCE/train_tensoIR.py" ] || { echo "no staged reference"; exit 4; }
nproc; python -c "import torch; print('torch threads', torch.get_num_threads())"
timeout -k 5 1500 python -m pytest tests/test_launcher.py -m gpu -q -rA --durations=5 > "$OUT/${TAG}_launcher_tests.log" 2>&1
echo "launcher tests rc=$?" | tee -a "$OUT/${TAG}_launcher_tests.log"
tail -12 "$OUT/${TAG}_launcher_tests.log"
timeout -k 5 1500 python oracle/ref_on_gpu.py --out "$OUT/${TAG}_ref_on_gpu.json" "$@" > "$OUT/${TAG}_ref_on_gpu.log" 2>&1
echo "ref_on_gpu rc=$?"
tail -c 3000 "$OUT/${TAG}_ref_on_gpu.log"
