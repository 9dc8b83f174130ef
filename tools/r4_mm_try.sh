#!/bin/bash
# round 4: MM concat step, autograd (eager / one hipGraph) against the one-call step, one box
R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2; do
for mode in eager graph step; do
  python3 tools/mm_profile.py concat 50000 200 $mode 2>&1 | tail -1
done
done
python3 tools/mm_profile.py concat 100000 200 step 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04_trace_mm_step -- python3 $R/tools/mm_profile.py concat 50000 30 step > $R/gpurun_out/r04_trace_mm_step.log 2>&1
python3 $R/tools/timeline.py $R/gpurun_out/r04_trace_mm_step "linear_nt_kernel<mmf::Tile<224" > $R/gpurun_out/r04_timeline_mm_step.txt
