"""Debug helper: exact-fp32 vs bf16x3 vs the fp64 oracle on one case; rows of dW1 beyond the bar and whether the oracle
itself shows a pre-activation of those units on the ReLU kink.   python tools/split_debug.py N size"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from multimodalfusion_amd import ops
from oracle import cases
from test_gpu_path import run_path_hip, relu_kink_units
N = int(sys.argv[1]); size = sys.argv[2]
m = dict(N=N, gated=True, size=size, K=8, dropout=True, y=6, c=1, alpha=0.25, bias_std=0.05, train=True, seed=8100, x_seed=8200, mask_seed=8300)
class MP:
    def setattr(self, obj, name, val): setattr(obj, name, val)
ref = cases.run_path(m)
sd, x, _ = cases.path_inputs(m)
u = np.asarray(x, np.float64) @ np.asarray(sd["attention_net_WSI.0.weight"], np.float64).T + np.asarray(sd["attention_net_WSI.0.bias"], np.float64)
for mode in (0, 1):
    ops.set_gemm(mode); r = run_path_hip(m, MP())
    k = "attention_net_WSI.0.weight"
    g = np.asarray(ref["grads"][k], np.float64); d = np.abs(r["grads"][k] - g)
    tol = 1e-5 + 1e-4 * np.abs(g).max()
    rows = np.unique(np.nonzero(d > tol)[0])
    print(f"mode {mode}: dW1 max err {d.max():.3e} (tol {tol:.3e}); rows beyond: {rows.tolist()}; min |u| of those units over the bag:",
          [float(np.abs(u[:, j]).min()) for j in rows])
print("units with min|u| < 4e-6:", sorted(relu_kink_units(sd, x)), " < 2e-5:", np.nonzero((np.abs(u) < 2e-5).any(axis=0))[0].tolist())
