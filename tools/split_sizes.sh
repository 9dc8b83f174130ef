#!/bin/bash
# environment overrides exist in the tuning build only (python tools/diag_build.py tune)
export MMF_LIB_PATH=${MMF_LIB_PATH:-multimodalfusion_amd/_diag/libmmf_tune.so}
# one-bag step time by bag size, exact fp32 vs bf16x3 with the small split tiles forced (MMF_SPLIT_MIN=1)
for n in ${SIZES:-1000 2000 4096 6000 10000 14000}; do
  for g in f32 bf16x3; do
    MMF_SPLIT_MIN=1 timeout -k 10 200 python bench.py --bag $n --gemm $g --steps 200 --warmup 20 --inflight 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$n', '$g', round(d['ms_per_step'],4), {k:round(v,1) for k,v in d['kernels_us'].items() if v>8})" || exit 1
  done
done
