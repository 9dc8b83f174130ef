#!/bin/bash
# phase split of the split-operand kernels: kernel averages of the 50k step under the s_* diagnostic builds
export MMF_GEMM=1
for v in ${VARIANTS:-normal s_nomfma s_nostage s_nosplit s_nogload s_noldsw}; do
  if [ "$v" = normal ]; then unset MMF_LIB_PATH; else export MMF_LIB_PATH=$PWD/multimodalfusion_amd/_diag/libmmf_$v.so; fi
  echo "== $v"
  timeout -k 10 200 python bench.py --steps 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k:round(v,1) for k,v in d['kernels_us'].items() if v>12})" || exit 1
done
