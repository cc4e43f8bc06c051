#!/bin/bash
# closing checks on the final library: smoke(), bench.py starting its own two ranks (gloo, one GPU) for every workload
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | tail -3
bash tools/two_rank_gloo.sh 2>&1 | tee gpurun_out/r04_two_rank_gloo.txt | tail -30
