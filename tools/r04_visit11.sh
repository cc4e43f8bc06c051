#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_parity.py -m gpu -q -x 2>&1 | grep -E "passed|failed" | tail -2
cat > /tmp/pb.py <<'PY'
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], {k:d[k] for k in ('value','ms_per_step')}, 'single', d['single_stream']['value'], [ (k['kernel'], round(k['avg_ms'],4)) for k in d['kernels'][:2]])
PY
timeout -k 5 300 python bench.py --no-side-workloads --no-sharp-scene --no-exact-pass --boundary-calls 3 --no-cpu-baseline 2>/dev/null | python /tmp/pb.py new
timeout -k 5 300 python bench.py --workload relight --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('relight', d['value'], d['ms_per_step'])"
