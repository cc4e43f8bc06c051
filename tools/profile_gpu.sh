#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of the bench command.
# Usage: tools/profile_gpu.sh <tag>     -> gpurun_out/prof_<tag>/...
set -u
TAG="${1:-r01}"
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$REPO/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 1 --in-flight 1 --no-sharp-scene --no-exact-pass --boundary-calls 3"
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $CMD > "$OUT/trace.log" 2>&1
timeout -k 5 300 rocprofv3 -L > "$OUT/counters.txt" 2>&1
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o fetch -- $CMD > "$OUT/pmc_fetch.log" 2>&1
timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o write -- $CMD > "$OUT/pmc_write.log" 2>&1
timeout -k 5 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d "$OUT/pmc_mfma" -o mfma -- $CMD > "$OUT/pmc_mfma.log" 2>&1
timeout -k 5 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/pmc_l2" -o l2 -- $CMD > "$OUT/pmc_l2.log" 2>&1
ls -R "$OUT" | head -50
python $REPO/tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# keep only small files
find "$OUT" -name "*.db" -size +20M -delete
