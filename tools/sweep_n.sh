#!/bin/bash
# environment overrides exist in the tuning build only (python tools/diag_build.py tune)
export MMF_LIB_PATH=${MMF_LIB_PATH:-multimodalfusion_amd/_diag/libmmf_tune.so}
# usage: tools/sweep_n.sh N "<ENV...>" ...  -> kernel us per env combo at bag size N (eval of tile heuristics)
N=$1; shift
for combo in "$@"; do
  echo "== N=$N $combo"
  env $combo timeout -k 10 120 python bench.py --bag $N --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_us']
print(round(d['value'],1), round(d['ms_per_step'],4), {n.replace('_kernel',''):round(v) for n,v in k.items() if v>9}, 'gemm_sum', round(sum(v for n,v in k.items() if n in ('linear_nt_kernel','gate_fwd_kernel','bwd_dh_kernel','tn_kernel'))))"
done
