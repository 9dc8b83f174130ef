"""Phase times of the omic one-launch step (workgroup 0's cycle stamps; needs the `stamps` diagnostic library:
python tools/diag_build.py stamps; MMF_LIB_PATH=multimodalfusion_amd/_diag/libmmf_stamps.so python tools/stamps_maxnet.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from multimodalfusion_amd import _lib, ops
from multimodalfusion_amd.models import MaxNet
dev = torch.device("cuda", 0)
m = MaxNet(input_dim=36, bag_loss="cox_surv").to(dev).train()
x = torch.randn(128, 36, device=dev); t = (torch.rand(128, dtype=torch.float64) * 100).to(dev); c = (torch.rand(128, device=dev) < 0.5).float()
params = [m.fc_omic[0][0].weight, m.fc_omic[0][0].bias, m.fc_omic[1][0].weight, m.fc_omic[1][0].bias, m.classifier.weight, m.classifier.bias]
grads = [torch.zeros_like(p) for p in params]
l = _lib.lib()
nbytes = l.mmf_maxnet_cox_step_workspace_bytes(128)
ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
risk = torch.empty(128, device=dev); loss = torch.empty((), device=dev)
sw = ops.sync_words(dev)
d = _lib.MaxnetDesc(B=128, G=36, H0=256, H1=256, x=x.data_ptr(), W0=params[0].data_ptr(), b0=params[1].data_ptr(), W1=params[2].data_ptr(),
                    b1=params[3].data_ptr(), Wc=params[4].data_ptr(), bc=params[5].data_ptr(), p_drop=0.25, seed=7, seed_dev=None,
                    sync=sw.data_ptr(), sync_words=ops.SYNC_WORDS, trace=None)
g = _lib.MaxnetGrads(*[q.data_ptr() for q in grads])
for it in range(5):
    rc = l.mmf_maxnet_cox_step(C.byref(d), t.data_ptr(), c.data_ptr(), 1.0, ws.data_ptr(), nbytes, risk.data_ptr(), loss.data_ptr(), C.byref(g), 0, None)
    assert rc == 0
    torch.cuda.synchronize()
    off = (4 * 128 * 256 + 128) * 4
    st = ws[off:off + 128].view(torch.int64).cpu().numpy()
    names = ["L0 gemm", "L1 + classifier", "barrier 1", "Cox", "phase 3", "barrier 2", "phase 4"]
    print("  ".join(f"{n} {int(st[i + 1] - st[i])}" for i, n in enumerate(names)), " total cycles", int(st[7] - st[0]),
          "| phase 4: staging", int(st[8] - st[6]), "dW1", int(st[9] - st[8]), "dW0", int(st[10] - st[9]), "db", int(st[11] - st[10]), "dWc/dbc", int(st[7] - st[11]),
          "| layer 1 from phase start: loads issued + pre", int(st[12] - st[1]), "first chunk in LDS", int(st[13] - st[12]), "chunk 0 multiplied", int(st[14] - st[13]),
          "chunks 1-3", int(st[15] - st[14]), "epilogue + classifier", int(st[2] - st[15]))
