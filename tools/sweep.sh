#!/bin/bash
# environment overrides exist in the tuning build only (python tools/diag_build.py tune)
export MMF_LIB_PATH=${MMF_LIB_PATH:-multimodalfusion_amd/_diag/libmmf_tune.so}
# usage: tools/sweep.sh "<ENV1=a ENV2=b>" "<...>" ...   -> one bench line (kernel us) per env combo
for combo in "$@"; do
  echo "== $combo"
  env $combo timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_us']
print(round(d['value'],1), round(d['ms_per_step'],4), {n.replace('_kernel',''):round(v) for n,v in k.items() if v>12})"
done
