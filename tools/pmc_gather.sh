CMD="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --profile-steps 1 --in-flight 1 --no-sharp-scene --no-exact-pass --boundary-calls 1"
cd $GRAFT_REPO_ROOT
PMC_TIMEOUT=200 bash tools/pmc_one.sh pmc_ta "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" $CMD > gpurun_out/pmc_ta.txt 2>&1
PMC_TIMEOUT=200 bash tools/pmc_one.sh pmc_tcp "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" $CMD > gpurun_out/pmc_tcp.txt 2>&1
PMC_TIMEOUT=200 bash tools/pmc_one.sh pmc_sq "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" $CMD > gpurun_out/pmc_sq.txt 2>&1
for f in ta tcp sq; do echo "=== $f"; grep -A9 -E "k_vm_app_mfma<12, true, false|k_march_secondary_lds|k_mlp_bf16<3, true, false" gpurun_out/pmc_$f.txt | head -40; done
