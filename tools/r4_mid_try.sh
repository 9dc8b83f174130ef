#!/bin/bash
# round 4: mid-size plan knobs (tuning build), all in one call
export MMF_LIB_PATH=multimodalfusion_amd/_diag/libmmf_tune.so
export NSWEEP_INFLIGHT=0
S="8192 10000 12288 14000"
echo "== base";                 python tools/nsweep.py $S
echo "== DH_SHORT_MAX=1024";    MMF_DH_SHORT_MAX=1024 python tools/nsweep.py $S
echo "== GATE_BIG_MIN=400";     MMF_GATE_BIG_MIN=400 python tools/nsweep.py $S
echo "== GATE_BIG_MIN=600";     MMF_GATE_BIG_MIN=600 python tools/nsweep.py $S
echo "== WIDE_MIN=8192 (wide tiles from 32 rows per CU)"; MMF_WIDE_MIN=8192 python tools/nsweep.py $S
echo "== TN_WIDE_MIN=8000";     MMF_TN_WIDE_MIN=8000 python tools/nsweep.py $S
