export TENSOIR_REFERENCE=$PWD/gpurun_scratch/reference
for v in auto:auto f16:auto auto:0 full:auto auto:auto; do
  p=${v%%:*}; dd=${v##*:}
  TENSOIR_INDIRECT_PRECISION=$p TENSOIR_DEVICE_DATASET=$dd python tools/script_head_to_head.py --out gpurun_out/v9_script_${p}_${dd}.json --modes hip --render 0 > /dev/null 2>&1
  python -c "import json; d=json.load(open('gpurun_out/v9_script_${p}_${dd}.json')); print('$v', d['hip']['ms_per_iteration'])"
done
# kernel trace of one default run (which launches fill the 400 iterations)
P="$PWD/gpurun_out/prof_v9_script"; mkdir -p "$P"; REPO=$PWD
T=$(mktemp -d); cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$REPO timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$P/trace" -o trace -- python -m tensoir_amd.run $TENSOIR_REFERENCE/train_tensoIR.py --config $REPO/tests/data/armadillo_compressed.txt --basedir $T > "$P/trace.log" 2>&1
cd $REPO; python tools/summarize_prof.py "$P/trace" > "$P/summary.txt" 2>&1; head -50 "$P/summary.txt"
find "$P" -name "*.db" -delete; find "$P" -name "*kernel_trace.csv" -size +3M -delete
