#!/bin/bash
# Round-3 profile set (run on the GPU box through gpurun; writes gpurun_out/r03_*):
#   kernel-trace stats of the one-bag step (fp32 50k) and of the bf16 100k step, then three PMC passes each
#   (FETCH_SIZE, WRITE_SIZE, MFMA busy) -- counters in their own runs, the program directly after `--` -- and the SQ wait /
#   LDS counters of the bf16 step (tools/f2_pmc.sh).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "50000 f32" "100000 bf16"; do
  set -- $cfg
  tag=$2_$1
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_trace_$tag -- python3 $R/tools/step_profile.py $1 100 $2 > $R/gpurun_out/r03_trace_$tag.log 2>&1 || exit 1
  python3 $R/tools/kstats.py $R/gpurun_out/r03_trace_$tag > $R/gpurun_out/r03_kstats_$tag.txt
  cp $(ls $R/gpurun_out/r03_trace_$tag/*/*kernel_stats.csv | head -1) $R/gpurun_out/r03_${tag}_kernel_stats.csv
  for ctr in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
    name=$(echo $ctr | tr ' ' '+')
    rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/r03_pmc_${tag}_$name -- python3 $R/tools/step_profile.py $1 12 $2 > $R/gpurun_out/r03_pmc_${tag}_$name.log 2>&1 || exit 1
  done
  python3 $R/tools/pmc_summary.py $R/gpurun_out/r03_pmc_${tag}_FETCH_SIZE $R/gpurun_out/r03_pmc_${tag}_WRITE_SIZE "$R/gpurun_out/r03_pmc_${tag}_SQ_VALU_MFMA_BUSY_CYCLES+SQ_BUSY_CYCLES" > $R/gpurun_out/r03_pmc_$tag.txt
  echo "done $tag"
done
cd $R && tools/f2_pmc.sh r03_sq > /dev/null 2>&1
echo "done sq"
