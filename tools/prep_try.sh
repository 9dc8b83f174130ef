#!/bin/bash
# environment overrides exist in the tuning build only (python tools/diag_build.py tune)
export MMF_LIB_PATH=${MMF_LIB_PATH:-multimodalfusion_amd/_diag/libmmf_tune.so}
# K-prep fused into the 64-row K-dh tiles up to which grid size?  (one bag per step, both gemm modes)
for n in 10000 14000; do
  for cap in 512 1024; do
    for g in f32 bf16x3; do
      MMF_DH_SHORT_MAX=$cap timeout -k 10 200 python bench.py --bag $n --gemm $g --steps 300 --warmup 30 --inflight 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$n', 'cap $cap', '$g', round(d['ms_per_step'],4), {k:round(v,1) for k,v in d['kernels_us'].items() if 'dh' in k or 'prep' in k})" || exit 1
    done
  done
done
