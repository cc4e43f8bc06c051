#!/bin/bash
# usage (GPU box): bash tools/two_rank_image_gloo.sh
# bench.py --workload image with two ranks on ONE GPU over gloo: plumbing test of the strong-scaling image path
# (row tiles + interleaved tiles, one all-gather per image) -- not a performance number
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29578 WORLD_SIZE=2 LOCAL_RANK=0
for TILE in 0 4096; do
  RANK=1 timeout 300 python bench.py --workload image --gpus 2 --backend gloo --allow-shared-gpu --steps 2 --warmup 1 --tile $TILE > gpurun_out/img_r1.out 2> gpurun_out/img_r1.err &
  RANK=0 timeout 300 python bench.py --workload image --gpus 2 --backend gloo --allow-shared-gpu --steps 2 --warmup 1 --tile $TILE > gpurun_out/img_r0.out 2> gpurun_out/img_r0.err
  rc0=$?; wait; echo "tile=$TILE rc0=$rc0"
  grep "^{\"metric" gpurun_out/img_r0.out | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_step','world_size','per_rank_render_ms','load_imbalance','exchange_ms')})"
  grep -i "error\|traceback" gpurun_out/img_r0.err gpurun_out/img_r1.err | head -5
done
