#!/bin/bash
export MMF_LIB_PATH=multimodalfusion_amd/_diag/libmmf_tune.so
export NSWEEP_INFLIGHT=0
for g in 256 512 1024 2048; do echo "== POOL_GROUPS=$g"; MMF_POOL_GROUPS=$g python tools/nsweep.py 10000 24000 50000; done
