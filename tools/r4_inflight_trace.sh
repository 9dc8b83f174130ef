#!/bin/bash
# round 4: kernel timeline of the two-bags-in-flight headline (who overlaps whom, where the GPU idles)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04_trace_inflight -- python3 $R/bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5 --blocks 2 > $R/gpurun_out/r04_trace_inflight.log 2>&1
ls $R/gpurun_out/r04_trace_inflight/*/ | head
