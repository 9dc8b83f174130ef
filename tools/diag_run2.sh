#!/bin/bash
ARGS="${ARGS:---dtype bf16 --bag 100000}"
for v in "$@"; do
  if [ "$v" = normal ]; then unset MMF_LIB_PATH; else export MMF_LIB_PATH=$PWD/multimodalfusion_amd/_diag/libmmf_$v.so; fi
  echo "== $v"
  timeout -k 10 200 python bench.py $ARGS --steps 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k:v for k,v in d['kernels_us'].items() if v>12})"
done
