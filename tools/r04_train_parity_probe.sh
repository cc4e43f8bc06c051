#!/bin/bash
cd $GRAFT_REPO_ROOT
cat > /tmp/pp.py <<'PY'
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['parity']
print(d['ms_per_step'], p['ok'], 'loss', p['loss_abs_diff'], 'dense', p['grad_max_rel'], 'field l2', p['field_grad_rel_l2'], 'outl', p['field_grad_outlier_share'], 'fmax', p['field_grad_max_rel'], 'maps', p['maps_max_abs'], list(p['worst_tensors'].items())[:2])
PY
for i in 1 2 3 4 5; do
    timeout 300 python bench.py --workload train 2>/dev/null | python /tmp/pp.py
done
