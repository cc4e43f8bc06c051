#!/bin/bash
# SQ counter passes for the march / gather / decoder kernels of the bench step (one lane)
cd $GRAFT_REPO_ROOT
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-side-workloads --profile-steps 1 --in-flight 1 --no-sharp-scene --no-exact-pass --boundary-calls 1 --sustained-steps 5"
PMC_TIMEOUT=200 bash tools/pmc_one.sh pmc_sq1 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS" $CMD > gpurun_out/pmc_sq1.txt 2>&1
PMC_TIMEOUT=200 bash tools/pmc_one.sh pmc_sq2 "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES" $CMD > gpurun_out/pmc_sq2.txt 2>&1
for f in sq1 sq2; do echo "=== $f"; grep -A9 -E "k_march_secondary_lds|k_indirect_fused|k_mlp_bf16_multi|k_vm_app_primary" gpurun_out/pmc_$f.txt | head -60; done
