#!/bin/bash
# round 4: the in-flight headline (two bags on two streams, distinct bags) for tile-height overrides, all in one call
export MMF_LIB_PATH=multimodalfusion_amd/_diag/libmmf_tune.so
run() { echo "== $*"; env "$@" python bench.py --no-extras --no-cpu-baseline --steps 100 --warmup 10 --blocks 6 --inflight ${INF:-2} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('  value', round(d['value'],1), 'one bag', round(d['value_one_bag_per_step'],1), {k.replace('_kernel',''):round(v) for k,v in d['kernels_us'].items() if v>20})"; }
run A=0
run MMF_WIDE_ROWS=208
run MMF_WIDE_ROWS=192
run MMF_WIDE_ROWS=240
run MMF_GATE_MIXED=2
run MMF_GATE_MIXED=0
INF=3 run A=0
run A=0
