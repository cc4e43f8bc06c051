#!/bin/bash
# dev visit: fused kernel variants (two-deep gather pipeline at 12 / 16 waves) -- A/B on one box + the fused-kernel tests on the 16-wave build
cd $GRAFT_REPO_ROOT; OUT=gpurun_out; mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-side-workloads --no-sharp-scene --no-exact-pass --boundary-calls 3 --sustained-steps 20"
one() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], d['ms_per_step'], 'single', d['single_stream']['ms_per_step'], ' '.join(f"{k['kernel'].replace('tir_','')}={k['avg_ms']:.4f}" for k in d['kernels'][:4]))
PY
}
for v in default p12 p16 default p16; do
  if [ $v = default ]; then $B > $OUT/v15_$v.json 2>$OUT/v15_$v.err; else TENSOIR_HIP_LIB=$PWD/gpurun_scratch/lib_$v.so $B > $OUT/v15_$v.json 2>$OUT/v15_$v.err; fi
  one $OUT/v15_$v.json
done
TENSOIR_HIP_LIB=$PWD/gpurun_scratch/lib_p16.so timeout -k 5 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "indirect_precision or renderer_boundary or mid_size or graph_replay" > $OUT/v15_tests.log 2>&1; echo "tests(p16) rc=$?"; tail -3 $OUT/v15_tests.log
