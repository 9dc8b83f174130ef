#!/bin/bash
# environment overrides exist in the tuning build only (python tools/diag_build.py tune)
export MMF_LIB_PATH=${MMF_LIB_PATH:-multimodalfusion_amd/_diag/libmmf_tune.so}
# 48- / 64-row wide tiles on bags below the usual wide-tile crossover: step time and parity
R=$GRAFT_REPO_ROOT
for n in 6000 10000 14000; do
for cfg in "" "MMF_WIDE_MIN=1" "MMF_WIDE_MIN=1 MMF_WIDE_ROWS=48" "MMF_WIDE_MIN=1 MMF_WIDE_ROWS=64"; do
  env $cfg timeout -k 10 120 python $R/bench.py --bag $n --steps 200 --warmup 20 --blocks 5 --no-extras --no-cpu-baseline --inflight 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('N', $n, '[$cfg]', round(d['ms_per_step'],4), {k: v for k, v in d['kernels_us'].items() if v > 12})"
done; done
MMF_WIDE_MIN=1 MMF_WIDE_ROWS=48 python -m pytest $R/tests/test_gpu_path.py $R/tests/test_gpu_nll_step.py -x -q -k "ragged or golden or step_matches or full_size" 2>&1 | tail -3
