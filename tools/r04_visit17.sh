#!/bin/bash
# dev visit: packed env-cell records in the relight integration -- pair-list test, C5 parity, relight bench, probe of kernel times
cd $GRAFT_REPO_ROOT; OUT=gpurun_out; mkdir -p $OUT
timeout -k 5 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_parity.py -m gpu -q -x -k "c5 or C5 or sampler or relight" > $OUT/v17_tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/v17_tests.log
timeout -k 5 300 python bench.py --workload relight --no-cpu-baseline > $OUT/v17_relight.json 2> $OUT/v17_relight.err; echo "relight rc=$?"
TENSOIR_ENV_RECORDS=0 timeout -k 5 300 python bench.py --workload relight --no-cpu-baseline > $OUT/v17_relight_tables.json 2> $OUT/v17_relight_tables.err; echo "relight(tables) rc=$?"
python - <<'PY'
import json
for f in ('v17_relight','v17_relight_tables'):
    d=json.loads(open(f'gpurun_out/{f}.json').read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
PY
