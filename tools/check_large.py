import sys, time, numpy as np, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from oracle import cases
from test_gpu_path import run_path_hip, compare
for N, size in ((100000, "small"), (30000, "big")):
    m = dict(N=N, gated=True, size=size, K=4, dropout=False, y=2, c=1, alpha=0.15, bias_std=0.02,
             train=False, seed=77, x_seed=78, mask_seed=0)
    t0 = time.time(); res = run_path_hip(m); t1 = time.time(); ref = cases.run_path(m); t2 = time.time()
    compare(res, ref, f"N={N} {size}")
    print(f"N={N} {size}: parity ok (hip {t1-t0:.1f}s incl. H2D, oracle fp64 {t2-t1:.1f}s)")
