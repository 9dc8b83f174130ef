#!/bin/bash
# Profile set of the bf16x3 (split-operand) step, fp32 50k bag (run on the GPU box through gpurun; writes gpurun_out/d_*):
# kernel-trace stats, then the PMC passes (HBM traffic; MFMA busy / cycles), each in its own run, the program directly after `--`.
R=$GRAFT_REPO_ROOT
export MMF_GEMM=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/d_trace -- python3 $R/tools/step_profile.py 50000 100 f32 > $R/gpurun_out/d_trace.log 2>&1 || exit 1
python3 $R/tools/kstats.py $R/gpurun_out/d_trace > $R/gpurun_out/d_kstats_bf16x3_50000.txt
for ctr in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo $ctr | tr ' ' '+')
  rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/d_pmc_$name -- python3 $R/tools/step_profile.py 50000 12 f32 > $R/gpurun_out/d_pmc_$name.log 2>&1 || exit 1
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/d_pmc_FETCH_SIZE $R/gpurun_out/d_pmc_WRITE_SIZE "$R/gpurun_out/d_pmc_SQ_VALU_MFMA_BUSY_CYCLES+SQ_BUSY_CYCLES+GRBM_GUI_ACTIVE" > $R/gpurun_out/d_pmc_bf16x3_50000.txt
cat $R/gpurun_out/d_kstats_bf16x3_50000.txt $R/gpurun_out/d_pmc_bf16x3_50000.txt
# the 10k bag (64-row split tiles, 128x128 split-K tile): kernel stats only
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/d_trace_10k -- python3 $R/tools/step_profile.py 10000 200 f32 > $R/gpurun_out/d_trace_10k.log 2>&1 || exit 1
python3 $R/tools/kstats.py $R/gpurun_out/d_trace_10k > $R/gpurun_out/d_kstats_bf16x3_10000.txt
cat $R/gpurun_out/d_kstats_bf16x3_10000.txt
