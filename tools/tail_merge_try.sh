#!/bin/bash
# environment overrides exist in the tuning build only (python tools/diag_build.py tune)
export MMF_LIB_PATH=${MMF_LIB_PATH:-multimodalfusion_amd/_diag/libmmf_tune.so}
for n in 1000 2000 4096; do
  for tm in 64 16 0; do
    MMF_TAIL_MERGE=$tm timeout -k 10 200 python bench.py --bag $n --steps 300 --warmup 30 --inflight 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$n', 'tail_merge<=$tm', round(d['ms_per_step'],4), {k:round(v,1) for k,v in d['kernels_us'].items() if 'tail' in k or 'merge' in k})" || exit 1
  done
done
