#!/bin/bash
# environment overrides exist in the tuning build only (python tools/diag_build.py tune)
export MMF_LIB_PATH=${MMF_LIB_PATH:-multimodalfusion_amd/_diag/libmmf_tune.so}
# gate_fwd tile plan (MMF_GATE_MIXED 0 / 1 / 2) under both gemm modes, one 50k bag per step
for g in bf16x3 f32; do
  for m in 1 2 0; do
    MMF_GATE_MIXED=$m timeout -k 10 200 python bench.py --gemm $g --steps 60 --warmup 10 --inflight 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$g', 'mixed=$m', round(d['ms_per_step'],4), {k:round(v,1) for k,v in d['kernels_us'].items() if 'gate' in k})" || exit 1
  done
done
