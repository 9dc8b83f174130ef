#!/bin/bash
# PMC pass over the bench (separate from --kernel-trace timing runs, as the guide prescribes).
# usage: tools/pmc.sh <outdir> <counters...>
out=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/$out.log 2>&1
echo "pmc exit=$?"
