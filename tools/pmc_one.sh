#!/bin/bash
# usage: tools/pmc_one.sh <outdir-under-gpurun_out> "<counters>" <python cmd...>   (GPU box)
# FETCH_SIZE and WRITE_SIZE (and the TCP_* sums) each need a pass of their own on gfx950: asking for two of them at once
# fails with "Request exceeds the capabilities of the hardware to collect" and rocprofv3 then hangs -- hence the timeout.
set -u
REPO="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$REPO/gpurun_out/$1"; shift
CTRS="$1"; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 5 ${PMC_TIMEOUT:-180} rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d "$OUT" -o pmc -- "$@" > "$OUT/log.txt" 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
files = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if k.startswith(("at::", "void at::", "__amd")): continue
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} mean={sum(v)/len(v):.4g} n={len(v)}")
PY
