#!/bin/bash
# round 4: the MM one-call step with / without the side stream over path-bag sizes (fp32 and bf16)
R=$GRAFT_REPO_ROOT
cd $R
for bf in 0 1; do
for n in 4000 10000 20000 50000 100000; do
for fm in 1000000000 1; do
  echo "bf16=$bf fork_min=$fm: $(MMF_MM_BF16=$bf MMF_MM_FORK_MIN=$fm python3 tools/mm_profile.py concat $n 200 step 2>&1 | tail -1)"
done
done
done
