#!/bin/bash
# dev visit: C5 pair lists (guided sampler) with per-kernel times; the pair-list and C5 parity tests
cd $GRAFT_REPO_ROOT; OUT=gpurun_out; mkdir -p $OUT
timeout -k 5 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_parity.py -m gpu -q -x -k "c5 or C5 or sampler or relight" > $OUT/v13_tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/v13_tests.log
timeout -k 5 300 python tools/c5_pairs_probe.py $OUT/c5_pairs_probe2.json > $OUT/v13_c5.log 2>&1; echo "c5 rc=$?"; grep pairs $OUT/v13_c5.log
timeout -k 5 300 python bench.py --workload relight --no-cpu-baseline > $OUT/v13_relight.json 2> $OUT/v13_relight.err; echo "relight rc=$?"; tail -c 1500 $OUT/v13_relight.json
