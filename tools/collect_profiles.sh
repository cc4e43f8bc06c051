#!/bin/bash
# Copy one GPU visit's evidence (tools/round_evidence.sh <tag> -> gpurun_out/) into the tracked profiles/ directory.
# usage: tools/collect_profiles.sh <tag>
set -u
TAG="${1:?tag}"
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; G="$REPO/gpurun_out"; P="$REPO/profiles"; D="$G/prof_$TAG"
for f in bench.json breakdown.json image_bench.json relight_bench.json train_bench.json parity_fullsize.json precision_policy_tests.json precision_trained_300.json; do
  [ -f "$G/${TAG}_$f" ] && cp "$G/${TAG}_$f" "$P/${TAG}_$f"
done
[ -f "$G/${TAG}_tests.log" ] && cp "$G/${TAG}_tests.log" "$P/${TAG}_gpu_tests.log"
cp "$D/summary.txt" "$P/${TAG}_rocprofv3_summary.txt"
cp "$D/summary_inflight2.txt" "$P/${TAG}_rocprofv3_summary_inflight2.txt"
cp "$D/summary_hp.txt" "$P/${TAG}_rocprofv3_summary_hp.txt" 2>/dev/null
cp "$D/summary_all.txt" "$P/${TAG}_rocprofv3_pmc_summary.txt"
for wl in image relight train; do cp "$D/summary_$wl.txt" "$P/${TAG}_rocprofv3_summary_$wl.txt"; done
for t in trace:"" trace2:_inflight2 trace_image:_image trace_relight:_relight trace_train:_train; do
  src="${t%%:*}"; suf="${t##*:}"
  f=$(find "$D/$src" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$P/${TAG}_kernel_stats$suf.csv"
done
cp "$D/pmc_traffic.json" "$P/${TAG}_pmc_traffic.json"; cp "$D/pmc_issue.json" "$P/${TAG}_pmc_issue.json"
# the two files bench.py reads (stamped with the library's source hash by tools/summarize_prof.py)
cp "$D/pmc_traffic.json" "$P/pmc_traffic.json"; cp "$D/pmc_issue.json" "$P/pmc_issue.json"
ls -la "$P" | grep "${TAG}_" | wc -l
