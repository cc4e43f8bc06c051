#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; P=$R/gpurun_out/prof_ta; mkdir -p $P
i=0
for pass in "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TD_TD_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" "TA_BUSY_avr TA_BUSY_max TA_TOTAL_WAVEFRONTS_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  timeout -k 5 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $P/p$i -o p$i -- python $R/tools/indirect_bench.py > $P/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python $R/tools/summarize_prof.py $P 2>&1 | grep -A200 "PMC" | grep -E "^#|k_indirect_fused|k_vm_app_mfma|k_mlp_bf16_auxt|k_march_secondary" | head -60
find $P -name "*.db" -delete
