#!/bin/bash
# usage (GPU box): bash tools/two_rank_gloo.sh
# two bench.py ranks on ONE GPU over gloo: plumbing test of the multi-rank flow (not a performance number)
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 WORLD_SIZE=2 LOCAL_RANK=0
RANK=1 timeout 250 python bench.py --gpus 2 --backend gloo --allow-shared-gpu --steps 6 --warmup 2 --no-exact-pass > gpurun_out/r1.out 2> gpurun_out/r1.err &
RANK=0 timeout 250 python bench.py --gpus 2 --backend gloo --allow-shared-gpu --steps 6 --warmup 2 --no-exact-pass > gpurun_out/r0.out 2> gpurun_out/r0.err
rc0=$?; wait; echo rc0=$rc0
grep "^{\"metric" gpurun_out/r0.out | cut -c1-260; echo "rank1 lines: $(grep -c metric gpurun_out/r1.out)"; grep -i "error\|traceback\|\[bench\]" gpurun_out/r0.err gpurun_out/r1.err | head -8
