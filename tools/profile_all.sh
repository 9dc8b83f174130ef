#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats (timing) and, in SEPARATE passes, the HBM traffic counters.
# usage (on the GPU box): bash tools/profile_all.sh <tag> [bench args...]     -> gpurun_out/<tag>_*
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $R/gpurun_out/${tag}_trace.log 2>&1
echo "trace exit=$?"
find $R/gpurun_out/${tag}_trace -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${tag}_kernel_stats.csv \;
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/${tag}_pmc_$c -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/${tag}_pmc_$c.log 2>&1
  echo "pmc $c exit=$?"
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/${tag}_pmc_FETCH_SIZE $R/gpurun_out/${tag}_pmc_WRITE_SIZE > $R/gpurun_out/${tag}_traffic.txt 2>&1
tail -20 $R/gpurun_out/${tag}_traffic.txt
