#!/bin/bash
# Round-4 profile set (run on the GPU box through gpurun; writes gpurun_out/r04_*):
#   kernel-trace stats of the one-bag step for fp32 1k / 10k / 50k and bf16 100k, then three PMC passes (FETCH_SIZE, WRITE_SIZE,
#   MFMA busy) for fp32 10k / 50k and bf16 100k -- counters in their own runs, the program directly after `--` -- and the
#   kernel-trace of BASELINE config 3 (radio one-call step, omic one-launch step) and of config 4 (the multimodal one-call step,
#   with one step's kernel timeline).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "1000 f32 400" "10000 f32 200" "50000 f32 100" "100000 bf16 100"; do
  set -- $cfg
  tag=$2_$1
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_trace_$tag -- python3 $R/tools/step_profile.py $1 $3 $2 > $R/gpurun_out/r04_trace_$tag.log 2>&1 || exit 1
  python3 $R/tools/kstats.py $R/gpurun_out/r04_trace_$tag > $R/gpurun_out/r04_kstats_$tag.txt
  cp $(ls $R/gpurun_out/r04_trace_$tag/*/*kernel_stats.csv | head -1) $R/gpurun_out/r04_${tag}_kernel_stats.csv
  echo "trace $tag done"
done
for cfg in "10000 f32" "50000 f32" "100000 bf16"; do
  set -- $cfg
  tag=$2_$1
  for ctr in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
    name=$(echo $ctr | tr ' ' '+')
    rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/r04_pmc_${tag}_$name -- python3 $R/tools/step_profile.py $1 12 $2 > $R/gpurun_out/r04_pmc_${tag}_$name.log 2>&1 || exit 1
  done
  python3 $R/tools/pmc_summary.py $R/gpurun_out/r04_pmc_${tag}_FETCH_SIZE $R/gpurun_out/r04_pmc_${tag}_WRITE_SIZE "$R/gpurun_out/r04_pmc_${tag}_SQ_VALU_MFMA_BUSY_CYCLES+SQ_BUSY_CYCLES" > $R/gpurun_out/r04_pmc_$tag.txt
  python3 $R/tools/pmc_to_traffic.py $1 $2 "profiles/r04/a_${tag}_pmc.txt" $R/gpurun_out/r04_pmc_${tag}_FETCH_SIZE $R/gpurun_out/r04_pmc_${tag}_WRITE_SIZE "$R/gpurun_out/r04_pmc_${tag}_SQ_VALU_MFMA_BUSY_CYCLES+SQ_BUSY_CYCLES" > $R/gpurun_out/r04_traffic_$tag.json
  echo "pmc $tag done"
done
for what in radio omic; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_trace_$what -- python3 $R/tools/radio_profile.py $what 100 > $R/gpurun_out/r04_trace_$what.log 2>&1 || exit 1
  python3 $R/tools/kstats.py $R/gpurun_out/r04_trace_$what > $R/gpurun_out/r04_kstats_$what.txt
done
echo "config 3 done"
# config 4: the multimodal concat step as the loop mirror runs it (MM_MIL_Attention_fc_surv.nll_step), kernel stats + one
# step's timeline (which kernels run beside which)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_trace_mm -- python3 $R/tools/mm_profile.py concat 50000 60 step > $R/gpurun_out/r04_trace_mm.log 2>&1 || exit 1
python3 $R/tools/kstats.py $R/gpurun_out/r04_trace_mm > $R/gpurun_out/r04_kstats_mm.txt
python3 $R/tools/timeline.py $R/gpurun_out/r04_trace_mm "linear_nt_kernel<mmf::Tile<224" > $R/gpurun_out/r04_timeline_mm.txt
echo "config 4 done"
