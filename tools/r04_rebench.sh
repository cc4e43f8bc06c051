#!/bin/bash
# re-run the bench lines only (after the PMC files of this library have been collected into profiles/): gpurun_out/<tag>_*
TAG="${1:-r04c}"; cd $GRAFT_REPO_ROOT; OUT=gpurun_out
timeout -k 5 900 python bench.py --breakdown "$OUT/${TAG}_breakdown.json" > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"; echo "bench rc=$?"
for wl in image relight train; do
  timeout -k 5 500 python bench.py --workload $wl > "$OUT/${TAG}_${wl}_bench.json" 2> "$OUT/${TAG}_${wl}_bench.err"; echo "$wl rc=$?"
done
