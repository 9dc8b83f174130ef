#!/bin/bash
# MFMA busy / shader cycles of the split-operand step (its own --pmc pass; the program directly after `--`)
R=$GRAFT_REPO_ROOT
export MMF_GEMM=${MMF_GEMM:-1}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/split_pmc
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/split_pmc -- python3 $R/tools/step_profile.py 50000 12 f32 > $R/gpurun_out/split_pmc.log 2>&1 || { tail -5 $R/gpurun_out/split_pmc.log; exit 1; }
python3 $R/tools/pmc_summary.py $R/gpurun_out/split_pmc | tee $R/gpurun_out/split_pmc.txt
