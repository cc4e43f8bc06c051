#!/bin/bash
# One GPU-box visit for the round's evidence: -m gpu tests, rocprofv3 kernel traces of the bench command
# (one batch in flight, then the default two), separate PMC passes (HBM traffic: FETCH_SIZE / WRITE_SIZE; issue: SQ counters),
# the three side workloads with their kernel traces, then the default bench line and the workload lines (with the fresh PMC files), and -- when a reference checkout is staged -- the bench line with the
# imported reference timed as cpu_baseline.  Usage (via gpurun): tools/r04_final.sh <tag>   -> gpurun_out/<tag>_*
set -u
TAG="${1:-r04}"
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$REPO/gpurun_out"
mkdir -p "$OUT"
cd "$REPO"
timeout -k 5 1200 python -m pytest tests -m gpu -q > "$OUT/${TAG}_tests.log" 2>&1
echo "tests rc=$?" >> "$OUT/${TAG}_tests.log"; tail -3 "$OUT/${TAG}_tests.log"
cp "$OUT/parity_fullsize.json" "$OUT/${TAG}_parity_fullsize.json" 2>/dev/null
P="$OUT/prof_${TAG}"; mkdir -p "$P"
cd /tmp && export TMPDIR=/tmp
CMD1="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 1 --boundary-calls 3 --no-exact-pass --no-sharp-scene --no-side-workloads --sustained-steps 10 --in-flight 1"
CMD2="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 1 --boundary-calls 3 --no-exact-pass --no-sharp-scene --no-side-workloads --sustained-steps 10"
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/trace" -o trace -- $CMD1 > "$P/trace.log" 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/trace2" -o trace -- $CMD2 > "$P/trace2.log" 2>&1
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "wait SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES"; do
  set -- $pass; name=$1; shift
  timeout -k 5 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$P/pmc_$name" -o $name -- $CMD1 > "$P/pmc_$name.log" 2>&1
  echo "pmc $name rc=$?"
done
for wl in image relight train; do
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/trace_$wl" -o trace -- python $REPO/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline > "$P/trace_$wl.log" 2>&1
  echo "trace $wl rc=$?"
done
# gpurun merges at most 64 MiB back: keep the stats / counter CSVs the summaries are made from, drop the databases and the
# per-dispatch traces of long runs
prune() { find "$P" -name "*.db" -delete; find "$P" -name "*kernel_trace.csv" -size +3M -delete; find "$P" -name "*counter_collection.csv" -size +8M -delete; find "$P" -name "*agent_info.csv" -delete; }
python "$REPO/tools/summarize_prof.py" "$P/trace" > "$P/summary.txt" 2>&1
python "$REPO/tools/summarize_prof.py" "$P/trace2" > "$P/summary_inflight2.txt" 2>&1
python "$REPO/tools/summarize_prof.py" "$P" > "$P/summary_all.txt" 2>&1
for wl in image relight train; do python "$REPO/tools/summarize_prof.py" "$P/trace_$wl" > "$P/summary_$wl.txt" 2>&1; done
cp "$P/pmc_traffic.json" "$OUT/${TAG}_pmc_traffic.json" 2>/dev/null; cp "$P/pmc_issue.json" "$OUT/${TAG}_pmc_issue.json" 2>/dev/null
# the bench lines come AFTER the counter passes: bench.py reads profiles/pmc_*.json, which must belong to this library
# (stamped with its source hash) -- on the box they are installed here, in the repository tools/collect_profiles.sh does it
if [ -s "$P/pmc_traffic.json" ] && [ -s "$P/pmc_issue.json" ]; then cp "$P/pmc_traffic.json" "$P/pmc_issue.json" "$REPO/profiles/"; fi
head -30 "$P/summary.txt"; tail -25 "$P/summary_all.txt"
prune; du -sh "$P" "$OUT"
cd "$REPO"
timeout -k 5 900 python bench.py --breakdown "$OUT/${TAG}_breakdown.json" > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
echo "bench rc=$?"; tail -c 600 "$OUT/${TAG}_bench.json"; echo; tail -3 "$OUT/${TAG}_bench.err"
for wl in image relight train; do
  timeout -k 5 500 python bench.py --workload $wl > "$OUT/${TAG}_${wl}_bench.json" 2> "$OUT/${TAG}_${wl}_bench.err"; echo "$wl rc=$?"; tail -c 300 "$OUT/${TAG}_${wl}_bench.json"; echo
done
if [ -f "$REPO/gpurun_scratch/reference/train_tensoIR.py" ]; then
  TENSOIR_REFERENCE="$REPO/gpurun_scratch/reference" timeout -k 5 600 python bench.py --no-sharp-scene --no-exact-pass --no-side-workloads > "$OUT/${TAG}_bench_refcpu.json" 2> "$OUT/${TAG}_bench_refcpu.err"
  echo "bench with reference cpu baseline rc=$?"; python -c "import json; d=json.load(open('$OUT/${TAG}_bench_refcpu.json')); print(d['cpu_baseline'], d.get('speedup_vs_cpu_baseline'))"
fi
prune; du -sh "$OUT"
