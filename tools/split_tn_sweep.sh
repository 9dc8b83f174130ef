#!/bin/bash
# environment overrides exist in the tuning build only (python tools/diag_build.py tune)
export MMF_LIB_PATH=${MMF_LIB_PATH:-multimodalfusion_amd/_diag/libmmf_tune.so}
# K-split balance of the split-operand TN kernel: plain (dW1) splits x gate (d[Wa;Wb]) splits, 4 s1 + 2 sg <= 256
for pair in "42 44" "40 48" "38 52" "36 56" "34 60"; do
  set -- $pair
  echo "== dW1 splits $1, gate splits $2"
  MMF_TN_SPLITS=$1 MMF_TN_GATE_SPLITS=$2 timeout -k 10 200 python bench.py --gemm bf16x3 --steps 30 --no-cpu-baseline --no-extras --inflight 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), {k:round(v,1) for k,v in d['kernels_us'].items() if v>12})" || exit 1
done
