#!/bin/bash
# round 4: kernel timeline of the MM concat step (config 4): which kernels overlap, where the path stack waits
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04_trace_mm -- python3 $R/tools/mm_profile.py ${1:-concat} 50000 30 > $R/gpurun_out/r04_trace_mm.log 2>&1
python3 $R/tools/timeline.py $R/gpurun_out/r04_trace_mm > $R/gpurun_out/r04_timeline_mm.txt
tail -3 $R/gpurun_out/r04_trace_mm.log
