#!/bin/bash
# LDS bank conflicts and VALU instruction counts of the bf16x3 step (its own --pmc pass; the program directly after `--`)
R=$GRAFT_REPO_ROOT
export MMF_GEMM=${MMF_GEMM:-1}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/d_pmc_lds
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/d_pmc_lds -- python3 $R/tools/step_profile.py 50000 12 f32 > $R/gpurun_out/d_pmc_lds.log 2>&1 || { tail -5 $R/gpurun_out/d_pmc_lds.log; exit 1; }
python3 $R/tools/pmc_summary.py $R/gpurun_out/d_pmc_lds | tee $R/gpurun_out/d_pmc_lds.txt
