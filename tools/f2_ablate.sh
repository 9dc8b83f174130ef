#!/bin/bash
# environment overrides exist in the tuning build only (python tools/diag_build.py tune)
export MMF_LIB_PATH=${MMF_LIB_PATH:-multimodalfusion_amd/_diag/libmmf_tune.so}
# kernel time of the bf16 fused forward with phases left out (diagnostic build f2dbg), two and one workgroup per CU
R=$GRAFT_REPO_ROOT
for l in 0 100000; do for m in "$@"; do
MMF_F2_LDS=$l MMF_F2_DEBUG_MASK=$m MMF_LIB_PATH=$R/multimodalfusion_amd/_diag/libmmf_f2dbg.so timeout -k 10 120 python $R/bench.py --dtype bf16 --bag 100000 --steps 30 --warmup 5 --no-extras --no-cpu-baseline --inflight 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('lds', $l, 'mask', $m, 'fused_us', d['kernels_us']['amil_fwd_fused_bf16_kernel'])"
done; done
