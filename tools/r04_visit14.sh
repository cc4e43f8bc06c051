#!/bin/bash
# dev visit: fused kernel with invariant-divisor index arithmetic -- precision-policy / parity tests, quick bench
cd $GRAFT_REPO_ROOT; OUT=gpurun_out; mkdir -p $OUT
timeout -k 5 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "indirect_precision or renderer_boundary or mid_size or graph_replay or c5_pair" > $OUT/v14_tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/v14_tests.log
timeout -k 5 300 python bench.py --no-cpu-baseline --no-side-workloads --no-sharp-scene --no-exact-pass > $OUT/v14_bench.json 2> $OUT/v14_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/v14_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('single_stream'))
for k in d['kernels'][:4]: print(k['kernel'], k['avg_ms'])
PY
