#!/bin/bash
# usage (GPU box): bash tools/two_rank_gloo.sh
# bench.py starting its OWN two ranks (bare `--gpus 2`, no launcher environment: bench.self_launch) on ONE GPU over gloo, every
# workload: plumbing test of the multi-rank flows (not a performance number; --allow-shared-gpu lifts the one-GPU-per-rank check
# for exactly this).  Wherever two GPUs exist, drop `--backend gloo --allow-shared-gpu` and the same command runs over RCCL.
for WL in batch image train relight; do
  EXTRA="--steps 4 --warmup 1 --no-exact-pass --no-sharp-scene --no-cpu-baseline --sustained-steps 8 --no-side-workloads"
  [ "$WL" = image ] && EXTRA="--steps 1 --warmup 0 --no-cpu-baseline"
  [ "$WL" = relight ] && EXTRA="--steps 1 --warmup 0 --no-cpu-baseline --maps 2"
  [ "$WL" = train ] && EXTRA="--steps 4 --warmup 1 --no-cpu-baseline"
  env -u WORLD_SIZE -u RANK -u LOCAL_RANK timeout 500 python bench.py --workload $WL --gpus 2 --backend gloo --allow-shared-gpu $EXTRA \
    > gpurun_out/two_rank_$WL.out 2> gpurun_out/two_rank_$WL.err
  echo "$WL rc=$?"
  grep "^{\"metric" gpurun_out/two_rank_$WL.out | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('value','ms_per_step','n_gpus','world_size','device_count','backend','per_rank_ms_per_step','per_rank_render_ms','scaling')})"
  grep -i "error\|traceback\|\[bench\]" gpurun_out/two_rank_$WL.err | head -6
done
# the launch check: a launcher environment that disagrees with --gpus is refused
WORLD_SIZE=1 python bench.py --gpus 2 --steps 1 2>&1 | tail -1
