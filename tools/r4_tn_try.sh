#!/bin/bash
export MMF_LIB_PATH=multimodalfusion_amd/_diag/libmmf_tune.so
export NSWEEP_INFLIGHT=0
for s in 0 8 10 12 16 21; do echo "== TN_SPLITS=$s"; MMF_TN_SPLITS=$s python tools/nsweep.py 6000 10000 14000 2>/dev/null; done
