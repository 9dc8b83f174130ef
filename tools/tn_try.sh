#!/bin/bash
# environment overrides exist in the tuning build only (python tools/diag_build.py tune)
export MMF_LIB_PATH=${MMF_LIB_PATH:-multimodalfusion_amd/_diag/libmmf_tune.so}
# fp32 headline with TN split / XCD-map overrides
for cfg in "0 0" "40 0" "40 1" "32 1" "48 1"; do
  set -- $cfg
  export MMF_TN_SPLITS=$1 MMF_TN_XCD=$2
  [ "$1" = 0 ] && unset MMF_TN_SPLITS
  echo "== splits=$1 xcd=$2"
  timeout -k 10 200 python bench.py --steps 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), {k:v for k,v in d['kernels_us'].items() if v>100})"
done
