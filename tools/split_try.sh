#!/bin/bash
# environment overrides exist in the tuning build only (python tools/diag_build.py tune)
export MMF_LIB_PATH=${MMF_LIB_PATH:-multimodalfusion_amd/_diag/libmmf_tune.so}
# A/B of the split-operand (bf16x3) GEMM path: parity tests with MMF_GEMM=1, then kernel stats of the 50k step.
R=$GRAFT_REPO_ROOT
export MMF_GEMM=1
cd $R && timeout -k 10 600 python -m pytest tests/test_gpu_path.py tests/test_gpu_nll_step.py -x -q -m gpu > $R/gpurun_out/split_tests.log 2>&1 || { tail -30 $R/gpurun_out/split_tests.log; exit 1; }
tail -3 $R/gpurun_out/split_tests.log
cd /tmp && export TMPDIR=/tmp
for rows in ${SPLIT_ROWS_LIST:-224}; do
  export MMF_SPLIT_ROWS=$rows
  rm -rf $R/gpurun_out/split_trace_$rows
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/split_trace_$rows -- python3 $R/tools/step_profile.py 50000 60 f32 > $R/gpurun_out/split_trace_$rows.log 2>&1 || exit 1
  python3 $R/tools/kstats.py $R/gpurun_out/split_trace_$rows > $R/gpurun_out/split_kstats_$rows.txt
  echo "rows $rows"; head -12 $R/gpurun_out/split_kstats_$rows.txt
done
