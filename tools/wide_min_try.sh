#!/bin/bash
# environment overrides exist in the tuning build only (python tools/diag_build.py tune)
export MMF_LIB_PATH=${MMF_LIB_PATH:-multimodalfusion_amd/_diag/libmmf_tune.so}
# from which bag size do the wide (224 x 256, one workgroup per CU) tiles beat the 64-row tiles?  both gemm modes
for n in 16384 24000 32768 40000; do
  for wm in 16384 65536; do
    for g in bf16x3 f32; do
      MMF_WIDE_MIN=$wm timeout -k 10 200 python bench.py --bag $n --gemm $g --steps 150 --warmup 20 --inflight 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$n', 'wide_min $wm', '$g', round(d['ms_per_step'],4))" || exit 1
    done
  done
done
