#!/bin/bash
# usage (GPU box): bash tools/two_rank_gloo.sh
# two bench.py ranks on ONE GPU over gloo, every workload: plumbing test of the multi-rank flows (not a performance number;
# --allow-shared-gpu lifts the one-GPU-per-rank check for exactly this)
export MASTER_ADDR=127.0.0.1 WORLD_SIZE=2 LOCAL_RANK=0
port=29577
for WL in batch train relight; do
  port=$((port + 1)); export MASTER_PORT=$port
  EXTRA="--steps 4 --warmup 1 --no-exact-pass --no-sharp-scene --no-cpu-baseline --sustained-steps 8"
  [ "$WL" = relight ] && EXTRA="--steps 1 --warmup 0 --no-cpu-baseline --maps 2"
  [ "$WL" = train ] && EXTRA="--steps 4 --warmup 1 --no-cpu-baseline"
  RANK=1 timeout 400 python bench.py --workload $WL --gpus 2 --backend gloo --allow-shared-gpu $EXTRA > gpurun_out/r1_$WL.out 2> gpurun_out/r1_$WL.err &
  RANK=0 timeout 400 python bench.py --workload $WL --gpus 2 --backend gloo --allow-shared-gpu $EXTRA > gpurun_out/r0_$WL.out 2> gpurun_out/r0_$WL.err
  rc0=$?; wait; echo "$WL rc0=$rc0"
  grep "^{\"metric" gpurun_out/r0_$WL.out | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('value','ms_per_step','n_gpus','world_size','device_count','backend','per_rank_ms_per_step','scaling')})"
  echo "rank1 lines: $(grep -c metric gpurun_out/r1_$WL.out)"; grep -i "error\|traceback\|\[bench\]" gpurun_out/r0_$WL.err gpurun_out/r1_$WL.err | head -6
done
# the launch check: a scaling line must not be printable from fewer ranks than it claims
python bench.py --gpus 2 --steps 1 2>&1 | tail -1
