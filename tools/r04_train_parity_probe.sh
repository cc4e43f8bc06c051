#!/bin/bash
cd $GRAFT_REPO_ROOT
cat > /tmp/pp.py <<'PY'
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['parity']
print(d['ms_per_step'], p['ok'], 'flips', p['record_mask_mismatches'], 'dense', p['grad_max_rel'], 'field l2', p['field_grad_rel_l2'], 'outl', p['field_grad_outlier_share'], 'fmax', p['field_grad_max_rel'], 'maps', max(p['maps_max_abs'].values()))
PY
for i in 1 2 3 4 5 6 7 8; do
    timeout 300 python bench.py --workload train --steps 60 --warmup 5 2>/dev/null | python /tmp/pp.py
done
