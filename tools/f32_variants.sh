#!/bin/bash
# kernel times of the fp32 50k one-bag step for library variants (tools/diag_build.py), in ONE gpurun call: usage f32_variants.sh <variant>...
R=$GRAFT_REPO_ROOT
for v in "$@"; do
L=$R/multimodalfusion_amd/_diag/libmmf_$v.so; [ "$v" = product ] && L=$R/multimodalfusion_amd/libmmf_amil.so
MMF_LIB_PATH=$L timeout -k 10 120 python $R/bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline --inflight 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_us']; print('$v', 'step_ms', round(d['ms_per_step'],4), {n: k[n] for n in ('tn_kernel','linear_nt_kernel','bwd_dh_kernel','gate_fwd_kernel')})"
done
