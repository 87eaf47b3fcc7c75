#!/bin/bash
# SQ counters of the consensus pair kernel on an N-run-heavy alignment (tools/nrun_bench.py, one case, AUTO only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CASE=${1:-0.05:0.5}
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $set | cut -d' ' -f2)
  rm -rf $R/gpurun_out/nrun_pmc_$tag
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/nrun_pmc_$tag -- python3 $R/tools/nrun_bench.py --cases $CASE --paths auto --reps 2 > $R/gpurun_out/nrun_pmc_$tag.log 2>&1 || { tail -5 $R/gpurun_out/nrun_pmc_$tag.log; exit 1; }
  grep -v amdgpu $R/gpurun_out/nrun_pmc_$tag.log | tail -1
done
