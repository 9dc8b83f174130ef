#!/bin/bash
# SQ counters of the bf16 100k step (LDS conflicts, wait breakdown, MFMA / VALU busy): two --pmc passes, each its own run,
# the program directly after `--`.   usage: tools/f2_pmc.sh <tag>
R=$GRAFT_REPO_ROOT
tag=${1:-f2}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/${tag}_pmc$i
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/${tag}_pmc$i -- python3 $R/tools/step_profile.py 100000 12 bf16 > $R/gpurun_out/${tag}_pmc$i.log 2>&1 || { tail -5 $R/gpurun_out/${tag}_pmc$i.log; exit 1; }
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/${tag}_pmc1 $R/gpurun_out/${tag}_pmc2 | tee $R/gpurun_out/${tag}_pmc.txt
