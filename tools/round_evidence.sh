#!/bin/bash
# A round's evidence in ONE GPU-box visit: -m gpu tests, rocprofv3 kernel traces of the bench command (one batch in flight, then the
# default two), separate PMC passes (HBM traffic: FETCH_SIZE / WRITE_SIZE; issue: SQ counters), the side workloads with their
# kernel traces and the simulated-rank scaling block, the indirect-light precision evidence (trained 300^3 checkpoint), then the
# default bench line and the workload lines (with the fresh PMC files installed).  Usage (via gpurun): tools/round_evidence.sh <tag>
# -> gpurun_out/<tag>_*; tools/collect_profiles.sh <tag> copies the result into profiles/.
# (Replaces the per-round r03_/r04_/r05_*.sh one-offs.  Nothing of the reference travels to the box: the launcher tests are
# skipped there and run in the build container, where the checkout exists.)
set -u
TAG="${1:-r06}"
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$REPO/gpurun_out"; mkdir -p "$OUT"; cd "$REPO"
timeout -k 5 1800 python -m pytest tests -m gpu -q > "$OUT/${TAG}_tests.log" 2>&1
echo "tests rc=$?" >> "$OUT/${TAG}_tests.log"; tail -4 "$OUT/${TAG}_tests.log"
cp "$OUT/parity_fullsize.json" "$OUT/${TAG}_parity_fullsize.json" 2>/dev/null
cp "$OUT/precision_policy_tests.json" "$OUT/${TAG}_precision_policy_tests.json" 2>/dev/null
P="$OUT/prof_${TAG}"; mkdir -p "$P"
cd /tmp && export TMPDIR=/tmp
CMD1="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 1 --boundary-calls 3 --no-exact-pass --no-full-pass --no-sharp-scene --no-side-workloads --sustained-steps 10 --in-flight 1"
CMD2="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 1 --boundary-calls 3 --no-exact-pass --no-full-pass --no-sharp-scene --no-side-workloads --sustained-steps 10"
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/trace" -o trace -- $CMD1 > "$P/trace.log" 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/trace2" -o trace -- $CMD2 > "$P/trace2.log" 2>&1
# the hp route (what a trained checkpoint runs): the same command with the indirect-light policy forced
TENSOIR_INDIRECT_PRECISION=hp timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/trace_hp" -o trace -- $CMD1 > "$P/trace_hp.log" 2>&1
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "wait SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES"; do
  set -- $pass; name=$1; shift
  timeout -k 5 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$P/pmc_$name" -o $name -- $CMD1 > "$P/pmc_$name.log" 2>&1
  echo "pmc $name rc=$?"
done
for wl in image relight train; do
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/trace_$wl" -o trace -- python $REPO/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline > "$P/trace_$wl.log" 2>&1
  echo "trace $wl rc=$?"
done
prune() { find "$P" -name "*.db" -delete; find "$P" -name "*kernel_trace.csv" -size +3M -delete; find "$P" -name "*counter_collection.csv" -size +8M -delete; find "$P" -name "*agent_info.csv" -delete; }
python "$REPO/tools/summarize_prof.py" "$P/trace" > "$P/summary.txt" 2>&1
python "$REPO/tools/summarize_prof.py" "$P/trace2" > "$P/summary_inflight2.txt" 2>&1
python "$REPO/tools/summarize_prof.py" "$P/trace_hp" > "$P/summary_hp.txt" 2>&1
python "$REPO/tools/summarize_prof.py" "$P" > "$P/summary_all.txt" 2>&1
for wl in image relight train; do python "$REPO/tools/summarize_prof.py" "$P/trace_$wl" > "$P/summary_$wl.txt" 2>&1; done
cp "$P/pmc_traffic.json" "$OUT/${TAG}_pmc_traffic.json" 2>/dev/null; cp "$P/pmc_issue.json" "$OUT/${TAG}_pmc_issue.json" 2>/dev/null
# the bench lines come AFTER the counter passes: bench.py reads profiles/pmc_*.json, which must belong to this library
if [ -s "$P/pmc_traffic.json" ] && [ -s "$P/pmc_issue.json" ]; then cp "$P/pmc_traffic.json" "$P/pmc_issue.json" "$REPO/profiles/"; fi
head -24 "$P/summary.txt"
prune; du -sh "$P" "$OUT"
cd "$REPO"
timeout -k 5 900 python tools/precision_300.py --out "$OUT/${TAG}_precision_trained_300.json" > "$OUT/${TAG}_precision.log" 2>&1; echo "precision rc=$?"
[ -s "$OUT/${TAG}_precision_trained_300.json" ] && cp "$OUT/${TAG}_precision_trained_300.json" "$REPO/profiles/r06_precision_trained_300.json"
timeout -k 5 900 python bench.py --breakdown "$OUT/${TAG}_breakdown.json" > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
echo "bench rc=$?"; tail -c 500 "$OUT/${TAG}_bench.json"; echo; tail -3 "$OUT/${TAG}_bench.err"
timeout -k 5 500 python bench.py --workload image --simulate-ranks 8 > "$OUT/${TAG}_image_bench.json" 2> "$OUT/${TAG}_image_bench.err"; echo "image rc=$?"
timeout -k 5 500 python bench.py --workload relight --simulate-ranks 8 > "$OUT/${TAG}_relight_bench.json" 2> "$OUT/${TAG}_relight_bench.err"; echo "relight rc=$?"
timeout -k 5 500 python bench.py --workload train > "$OUT/${TAG}_train_bench.json" 2> "$OUT/${TAG}_train_bench.err"; echo "train rc=$?"
for wl in image relight train; do tail -c 200 "$OUT/${TAG}_${wl}_bench.json"; echo; done
prune; du -sh "$OUT"
