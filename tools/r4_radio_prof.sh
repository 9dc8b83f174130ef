#!/bin/bash
# round 4: kernel-trace of the radio / omic / mm steps (config 3 / 4)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for what in radio omic; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4_prof_$what -- python3 $R/tools/radio_profile.py $what 100 > $R/gpurun_out/r4_prof_$what.log 2>&1
  python3 $R/tools/kstats.py $R/gpurun_out/r4_prof_$what > $R/gpurun_out/r4_kstats_$what.txt
done
