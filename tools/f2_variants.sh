#!/bin/bash
# kernel time of the bf16 fused forward in compile-time diagnostic variants (tools/diag_build.py): usage f2_variants.sh <variant>...
R=$GRAFT_REPO_ROOT
for v in "$@"; do
L=$R/multimodalfusion_amd/_diag/libmmf_$v.so; [ "$v" = product ] && L=$R/multimodalfusion_amd/libmmf_amil.so
MMF_LIB_PATH=$L timeout -k 10 120 python $R/bench.py --dtype bf16 --bag 100000 --steps 30 --warmup 5 --no-extras --no-cpu-baseline --inflight 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', 'fused_us', d['kernels_us']['amil_fwd_fused_bf16_kernel'], 'dh_us', d['kernels_us']['dh_bf16_kernel'], 'tn_us', d['kernels_us']['tn_bf16_kernel'], 'step_ms', round(d['ms_per_step'],4))"
done
