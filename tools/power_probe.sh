#!/bin/bash
# Socket power and shader clock while the step loop runs (read-only rocm-smi queries): is a mode power-limited?
# usage: MMF_GEMM=0|1 bash tools/power_probe.sh
R=${GRAFT_REPO_ROOT:-$PWD}
# a BOUNDED run (about 10 s at 0.55-0.8 ms per step) that ends by itself: nothing is killed with kernels in flight
python3 $R/tools/step_profile.py 50000 ${STEPS:-14000} ${DTYPE:-f32} > /dev/null 2>&1 &
PID=$!
sleep 4
for i in 1 2 3 4; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | tr '\n' ' '; echo
  sleep 1
done
wait $PID
true
