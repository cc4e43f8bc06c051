#!/bin/bash
# dev visit: lane-parallel march set-up + C5 pair lists -- targeted tests, C5 mode probe, quick bench
cd $GRAFT_REPO_ROOT; OUT=gpurun_out; mkdir -p $OUT
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_parity.py -m gpu -q -x -k "lds_staged or c5 or C5 or occupied_box or edge_cases or overflow or hdr_relight or full_size or mid_size or graph or sampler or relight" > $OUT/v12_tests.log 2>&1
echo "tests rc=$?"; tail -5 $OUT/v12_tests.log
timeout -k 5 300 python tools/c5_pairs_probe.py $OUT/c5_pairs_probe.json > $OUT/v12_c5.log 2>&1; echo "c5 rc=$?"; cat $OUT/v12_c5.log | tail -12
timeout -k 5 300 python bench.py --no-cpu-baseline --no-side-workloads --no-sharp-scene --no-exact-pass > $OUT/v12_bench.json 2> $OUT/v12_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/v12_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('single_stream'), d['parity'].get('ok'))
for k in d['kernels'][:4]: print(k['kernel'], k['avg_ms'])
PY
