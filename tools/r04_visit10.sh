#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "precision_policy or graph or renderer_boundary or c4" 2>&1 | tail -3
cat > /tmp/pb.py <<'PY'
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], {k:d[k] for k in ('value','ms_per_step')}, 'single', d['single_stream']['value'], 'parity', d['parity']['max_abs'] if d.get('parity') else None, [ (k['kernel'], round(k['avg_ms'],4)) for k in d['kernels'][:2]])
PY
for v in base w8 base w8; do
  L=$GRAFT_REPO_ROOT/tensoir_amd/libtensoir_hip.so; [ $v != base ] && L=$GRAFT_REPO_ROOT/gpurun_scratch/lib_$v.so
  TENSOIR_HIP_LIB=$L timeout -k 5 300 python bench.py --no-side-workloads --no-sharp-scene --no-exact-pass --boundary-calls 3 --cpu-rays 256 --cpu-calls 1 2>/dev/null | python /tmp/pb.py $v
done
