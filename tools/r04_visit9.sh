#!/bin/bash
cd $GRAFT_REPO_ROOT
cat > /tmp/pb.py <<'PY'
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], {k:d[k] for k in ('value','ms_per_step')}, 'sustained', d['sustained']['value'], 'single', (d['single_stream'] or {}).get('value'))
PY
for nf in 1 2 3 4; do
  timeout -k 5 300 python bench.py --in-flight $nf --no-side-workloads --no-sharp-scene --no-exact-pass --boundary-calls 3 --no-cpu-baseline 2>/dev/null | python /tmp/pb.py lanes=$nf
done
timeout -k 5 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
