"""Debug: in a training-type forward (eval mode, so no dropout), are the scores consistent with the saved a / b, and are the
saved a / b consistent with the saved h?  Tells a phase-2 MFMA problem from a score-reduction problem."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalfusion_amd.models import MIL_Attention_fc_surv_path
DEV = torch.device("cuda", 0)
torch.manual_seed(5)
N = 100_000
model = MIL_Attention_fc_surv_path(gate_path=True, model_size_wsi="small", dropout=False, n_classes=4).to(DEV).eval()
x = torch.randn(N, 1024, device=DEV).to(torch.bfloat16)
H_OFF = 1024 + 1024 + 524288 + 262144 + 262144
A_OFF = H_OFF + N * 512 + 800000 + 807168 + 256 + 8448
B_OFF = A_OFF + N * 512
sd = {k: v.detach() for k, v in model.state_dict().items()}
Wa, ba = sd["attention_net_WSI.3.attention_a.0.weight"], sd["attention_net_WSI.3.attention_a.0.bias"]
Wb, bb = sd["attention_net_WSI.3.attention_b.0.weight"], sd["attention_net_WSI.3.attention_b.0.bias"]
Wc, bc = sd["attention_net_WSI.3.attention_c.weight"], sd["attention_net_WSI.3.attention_c.bias"]
bf = lambda t: t.to(torch.bfloat16).float()
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    hz, S, Yh, A = model(path_features=x)
    node = hz.grad_fn
    seen = set()
    stack = [node]
    ws = None
    while stack:
        n = stack.pop()
        if n is None or id(n) in seen:
            continue
        seen.add(id(n))
        if hasattr(n, "ws"):
            ws = n.ws
            break
        stack.extend(f for f, _ in n.next_functions)
    torch.cuda.synchronize()
    view = lambda off: ws[off:off + N * 512].view(torch.bfloat16).view(N, 256).float()
    h, a, b = view(H_OFF), view(A_OFF), view(B_OFF)
    s_ab = (a * b) @ Wc.flatten() + bc
    dA = (s_ab - A.flatten()).abs()
    bad = (dA > 1e-3).nonzero().flatten()
    a_ref = bf(torch.tanh(h @ bf(Wa).T + ba)); b_ref = bf(torch.sigmoid(h @ bf(Wb).T + bb))
    da = (a - a_ref).abs().amax(1); db = (b - b_ref).abs().amax(1)
    bad_ab = ((da > 0.02) | (db > 0.02)).nonzero().flatten()
    print(f"run {it}: scores vs saved a,b: {bad.numel()} rows off (max {dA.max().item():.3g}) rows {bad[:6].tolist()} in-tile {[int(v) % 128 for v in bad[:6]]};  "
          f"saved a/b vs h: {bad_ab.numel()} rows off (max {max(da.max().item(), db.max().item()):.3g}) rows {bad_ab[:6].tolist()}")
    if bad_ab.numel():
        r = int(bad_ab[0])
        cols = ((a[r] - a_ref[r]).abs() > 0.02).nonzero().flatten()
        print("   first bad a/b row", r, "in-tile", r % 128, "bad a dims:", cols[:40].tolist())
