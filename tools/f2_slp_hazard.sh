#!/bin/bash
# round 4: does the round-3 corruption of the fused bf16 forward (built with SLP vectorisation: packed-fp32 VALU) still
# show, and which single change removes it?  Each library: repeated forward-only scores (200 launches, the round-3
# reproducer tools/f2_debug.py), then 60 repeated training steps, outputs compared bit for bit.
for v in ${VARIANTS:-product f2slp f2slp_nopk f2slp_wz f2cond f2cond_slp}; do
  if [ $v = product ]; then lib=multimodalfusion_amd/libmmf_amil.so; else lib=multimodalfusion_amd/_diag/libmmf_$v.so; fi
  echo "== $v"
  MMF_LIB_PATH=$lib timeout -k 10 300 python tools/f2_debug.py 100000 200 2>&1 | grep -v amdgpu.ids | tail -5
  MMF_LIB_PATH=$lib timeout -k 10 300 python tools/bf16_determinism.py 100000 60 2>&1 | grep -v amdgpu.ids | tail -3
done
