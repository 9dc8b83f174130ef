#!/bin/bash
# bf16x3 mode (MMF_GEMM=1) of the fp32 50k step: kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes (each its own run, the
# program directly after `--`; the mode comes from the environment of this shell).  Writes gpurun_out/r03x3_*.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export MMF_GEMM=1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03x3_trace -- python3 $R/tools/step_profile.py 50000 100 f32 > $R/gpurun_out/r03x3_trace.log 2>&1 || exit 1
python3 $R/tools/kstats.py $R/gpurun_out/r03x3_trace > $R/gpurun_out/r03x3_kstats.txt
cp $(ls $R/gpurun_out/r03x3_trace/*/*kernel_stats.csv | head -1) $R/gpurun_out/r03x3_kernel_stats.csv
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/r03x3_pmc_$ctr -- python3 $R/tools/step_profile.py 50000 12 f32 > $R/gpurun_out/r03x3_pmc_$ctr.log 2>&1 || exit 1
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/r03x3_pmc_FETCH_SIZE $R/gpurun_out/r03x3_pmc_WRITE_SIZE > $R/gpurun_out/r03x3_pmc.txt
echo "done x3"
