"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Generates tests/golden/*.npz by importing the reference's own Python modules
from $MMF_REFERENCE (default /root/reference) on CPU and running them on inputs
from the repo-owned generator (oracle/inputs.py).  The reference never travels
to the GPU box; only the small outputs written here do.  Run:

    python -m oracle.gen_golden            # from the repo root, build container only

Shims recorded in every fixture's `meta`:
  * empty `torchvision` / `torchvision.transforms` modules (imported, never used, by
    utils/utils.py:10 and utils/loss_utils.py:10 of the reference);
  * MM model only: module global `size_path`, construction through the base-class
    __init__ (the subclass passes an unknown kwarg), 1-D genomic input, and for
    fusion='tensor' `torch.cuda.FloatTensor` mapped to the CPU constructor;
  * train-mode cases: `nn.Dropout.forward` / `nn.AlphaDropout.forward` replaced by a
    multiply with a mask from oracle.inputs.keep_mask, so the mask is known.
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch

from . import inputs as gen

REF = os.environ.get("MMF_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

N_SAMPLE = 256


def _import_reference():
    if not os.path.isdir(REF):
        raise SystemExit(f"reference not present at {REF}: nothing to do")
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tv.transforms = tvt
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tvt)
    sys.path.insert(0, REF)
    import models.model_attention_mil_path as mp
    import models.model_attention_mil_radio as mr
    import models.model_genomic as mg
    import models.model_mm_attention_mil as mm
    import models.model_modules as mods
    import utils.loss_utils as lu
    import utils.utils as uu
    return mp, mr, mg, mm, mods, lu, uu


def sample_idx(n: int, k: int = N_SAMPLE, salt: int = 0) -> np.ndarray:
    """Deterministic sample positions (shared with the tests)."""
    if n <= k:
        return np.arange(n, dtype=np.int64)
    u = gen.uniform01(977 + salt, k, stream=n % 65521)
    return np.unique((u * n).astype(np.int64))


def summarize(prefix: str, arr: np.ndarray, out: dict, full_below: int = 4096):
    a = np.asarray(arr, dtype=np.float64).reshape(-1)
    out[prefix + "/sum"] = np.float64(a.sum())
    out[prefix + "/l2"] = np.float64(np.sqrt((a * a).sum()))
    out[prefix + "/absmax"] = np.float64(np.abs(a).max()) if a.size else np.float64(0)
    if a.size <= full_below:
        out[prefix + "/full"] = a.copy()
    else:
        out[prefix + "/sample"] = a[sample_idx(a.size)]


def _load_sd(model, sd_np, dtype):
    sd = {k: torch.as_tensor(v).to(dtype) for k, v in sd_np.items()}
    model.load_state_dict(sd, strict=True)
    return model.to(dtype)


def _record(out, tag, hz, S, Yh, loss, model, A_raw=None, M=None):
    out[f"{tag}/hazards"] = hz.detach().double().numpy()
    if S is not None:
        out[f"{tag}/S"] = S.detach().double().numpy()
    if Yh is not None:
        out[f"{tag}/Y_hat"] = Yh.detach().numpy()
    out[f"{tag}/loss"] = np.float64(loss.item())
    if M is not None:
        out[f"{tag}/M"] = M.detach().double().numpy()
    if A_raw is not None:
        items = A_raw.items() if isinstance(A_raw, dict) else [("", A_raw)]
        for name, A in items:
            a = A.detach().double().numpy().reshape(-1)
            key = f"{tag}/A_raw{('_' + name) if name else ''}"
            summarize(key, a, out, full_below=1024)
            out[key + "/max"] = np.float64(a.max())
            out[key + "/argmax"] = np.int64(a.argmax())
            out[key + "/logsumexp"] = np.float64(np.log(np.exp(a - a.max()).sum()) + a.max())
    for k, p in model.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        summarize(f"{tag}/grad/{k}", g.detach().double().numpy(), out)


class _MaskQueue:
    """Replaces nn.Dropout / nn.AlphaDropout forward by a multiply with queued masks."""

    def __init__(self):
        self.queue = []
        self._orig = (torch.nn.Dropout.forward, torch.nn.AlphaDropout.forward)

    def __enter__(self):
        q = self

        def drop_fwd(mod, x):
            if not mod.training:
                return x
            m = q.queue.pop(0)
            assert m.shape == x.shape, (m.shape, x.shape)
            return x * m.to(x.dtype)

        def adrop_fwd(mod, x):
            if not mod.training:
                return x
            from .torch_port import alpha_dropout_apply
            keep = q.queue.pop(0)
            assert keep.shape == x.shape
            return alpha_dropout_apply(x, keep.to(x.dtype), mod.p)

        torch.nn.Dropout.forward = drop_fwd
        torch.nn.AlphaDropout.forward = adrop_fwd
        return self

    def __exit__(self, *a):
        torch.nn.Dropout.forward, torch.nn.AlphaDropout.forward = self._orig


# ----------------------------------------------------------------------------------
PATH_CASES = [
    # name, N, gated, size, K, dropout(ctor flag), y, c, alpha, bias_std, train
    ("g_small_k4_n1000", 1000, True, "small", 4, False, 1, 0, 0.0, 0.0, False),
    ("g_small_k4_n1000_b", 1000, True, "small", 4, False, 2, 1, 0.0, 0.05, False),
    ("g_small_k4_n1", 1, True, "small", 4, False, 0, 0, 0.6, 0.05, False),
    ("g_small_k4_n7", 7, True, "small", 4, False, 3, 1, 0.6, 0.05, False),
    ("g_small_k8_n1000", 1000, True, "small", 8, False, 7, 1, 0.6, 0.05, False),
    ("g_small_k8_n333", 333, True, "small", 8, False, 2, 0, 0.15, 0.05, False),
    ("g_big_k4_n1000", 1000, True, "big", 4, False, 2, 0, 0.0, 0.05, False),
    ("g_big_k8_n130", 130, True, "big", 8, True, 0, 1, 0.6, 0.05, False),
    ("u_small_k4_n1000", 1000, False, "small", 4, False, 2, 0, 0.0, 0.05, False),
    ("u_small_k8_n257_do", 257, False, "small", 8, True, 5, 1, 0.15, 0.05, False),
    ("u_big_k4_n64", 64, False, "big", 4, False, 3, 0, 0.6, 0.05, False),
    ("g_small_k4_n10000", 10000, True, "small", 4, False, 1, 0, 0.0, 0.0, False),
    ("g_small_k4_n300_train", 300, True, "small", 4, True, 2, 0, 0.0, 0.05, True),
    ("g_small_k4_n300_train1", 300, True, "small", 4, False, 1, 1, 0.15, 0.05, True),
    ("u_small_k4_n300_train", 300, False, "small", 4, True, 1, 0, 0.0, 0.05, True),
]


def gen_path(mp, lu):
    out = {}
    meta = {}
    for (name, N, gated, size, K, dropout, y, c, alpha, bias_std, train) in PATH_CASES:
        seed = 100 + len(meta)
        sd_np = gen.path_state_dict(seed=seed, gated=gated, size=size, n_classes=K,
                                    dropout=dropout, bias_std=bias_std)
        x_np = gen.bag(seed + 5000, N)
        meta[name] = dict(N=N, gated=gated, size=size, K=K, dropout=dropout, y=y, c=c, alpha=alpha,
                          bias_std=bias_std, train=train, seed=seed, x_seed=seed + 5000, mask_seed=seed + 9000)
        for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            model = mp.MIL_Attention_fc_surv_path(gate_path=gated, model_size_wsi=size,
                                                  dropout=dropout, n_classes=K)
            _load_sd(model, sd_np, dt)
            x = torch.as_tensor(x_np).to(dt)
            loss_fn = lu.NLLSurvLoss(alpha=alpha)
            H, D = gen.SIZE_DICT[size][1:]
            with _MaskQueue() as mq:
                if train:
                    model.train()
                    mq.queue.append(torch.as_tensor(gen.drop_scale_mask(seed + 9000, 0, N, H, 0.25, np.float64)))
                    if dropout:
                        mq.queue.append(torch.as_tensor(gen.drop_scale_mask(seed + 9000, 1, N, D, 0.25, np.float64)))
                        if gated:
                            mq.queue.append(torch.as_tensor(gen.drop_scale_mask(seed + 9000, 2, N, D, 0.25, np.float64)))
                else:
                    model.eval()
                hz, S, Yh, A_raw = model(path_features=x)
                M = model(path_features=x, return_features=True) if not train else None
                assert not mq.queue
            loss = loss_fn(hazards=hz, S=S, Y=torch.tensor([y]), c=torch.tensor([float(c)]))
            loss.backward()
            _record(out, f"{name}/{tag}", hz, S, Yh, loss, model, A_raw, M)
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "path.npz"), **out)
    print("path.npz", len(out), "entries")


RADIO_CASES = [
    # name, n, n_mod, gated, dropout, K, y, c, alpha, train
    ("m4_n512", 512, 4, True, True, 4, 1, 0, 0.0, False),
    ("m4_n77_u", 77, 4, False, False, 8, 6, 1, 0.15, False),
    ("m1_n512", 512, 1, True, True, 4, 3, 1, 0.0, False),
    ("m2_n100_train", 100, 2, True, True, 4, 0, 0, 0.0, True),
]
MODS = ["T1", "T2", "T1Gd", "FLAIR"]


def gen_radio(mr, lu):
    out, meta = {}, {}
    for (name, n, n_mod, gated, dropout, K, y, c, alpha, train) in RADIO_CASES:
        seed = 300 + len(meta)
        sd_np = gen.radio_state_dict(seed=seed, gated=gated, n_classes=K, dropout=dropout,
                                     n_mod=n_mod, bias_std=0.05)
        xs_np = [gen.bag(seed + 5000, n, stream=7 * i) for i in range(n_mod)]
        meta[name] = dict(n=n, n_mod=n_mod, gated=gated, dropout=dropout, K=K, y=y, c=c, alpha=alpha,
                          train=train, seed=seed, x_seed=seed + 5000, mask_seed=seed + 9000, bias_std=0.05)
        for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            model = mr.MIL_Attention_fc_surv_radio(radio_fusion="concat", gate_radio=gated, dropout=dropout,
                                                   n_classes=K, modalities=MODS[:n_mod])
            _load_sd(model, sd_np, dt)
            kw = {m: torch.as_tensor(x).to(dt) for m, x in zip(MODS, xs_np)}
            with _MaskQueue() as mq:
                if train:
                    model.train()
                    mq.queue.append(torch.as_tensor(gen.drop_scale_mask(seed + 9000, 0, n, 256, 0.25, np.float64)))
                    if dropout:
                        mq.queue.append(torch.as_tensor(gen.drop_scale_mask(seed + 9000, 1, n, 256, 0.25, np.float64)))
                        if gated:
                            mq.queue.append(torch.as_tensor(gen.drop_scale_mask(seed + 9000, 2, n, 256, 0.25, np.float64)))
                else:
                    model.eval()
                hz, S, Yh, A_raw = model(**kw)
                M = model(return_features=True, **kw) if not train else None
            loss = lu.NLLSurvLoss(alpha=alpha)(hazards=hz, S=S, Y=torch.tensor([y]), c=torch.tensor([float(c)]))
            loss.backward()
            _record(out, f"{name}/{tag}", hz, S, Yh, loss, model, A_raw, M)
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "radio.npz"), **out)
    print("radio.npz", len(out), "entries")


def omic_batch(seed, B, G, ties=True):
    x = gen.normal(seed, (B, G), stream=0)
    t = np.floor(gen.uniform01(seed, B, stream=11) * 100.0 * (0.25 if ties else 1000.0)) / (0.25 if ties else 1000.0)
    c = (gen.uniform01(seed, B, stream=12) < 0.5).astype(np.float32)
    return x, t.astype(np.float64), c


OMIC_CASES = [
    # name, B, G, nll, K, train
    ("cox_g36_b128", 128, 36, False, 4, False),
    ("cox_g186_b64_train", 64, 186, False, 4, True),
    ("cox_g36_b1", 1, 36, False, 4, False),
    ("nll_g36_b1", 1, 36, True, 4, False),
    ("nll_g186_b1_k8", 1, 186, True, 8, False),
]


def gen_omic(mg, lu):
    out, meta = {}, {}
    lu.device = torch.device("cpu")
    for (name, B, G, nll, K, train) in OMIC_CASES:
        seed = 500 + len(meta)
        sd_np = gen.maxnet_state_dict(seed=seed, input_dim=G, nll=nll, n_classes=K, bias_std=0.05)
        x_np, t_np, c_np = omic_batch(seed + 5000, B, G)
        y = 1
        meta[name] = dict(B=B, G=G, nll=nll, K=K, train=train, seed=seed, x_seed=seed + 5000,
                          mask_seed=seed + 9000, y=y, alpha=0.15, bias_std=0.05)
        for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            model = mg.MaxNet(input_dim=G, model_size_omic="small",
                              bag_loss="nll_surv" if nll else "cox_surv", n_classes=K)
            _load_sd(model, sd_np, dt)
            x = torch.as_tensor(x_np).to(dt)
            with _MaskQueue() as mq:
                if train:
                    model.train()
                    for i in range(2):
                        mq.queue.append(torch.as_tensor(gen.keep_mask(seed + 9000, i, B, 256, 0.25).astype(np.float64)))
                else:
                    model.eval()
                res = model(genomic_features=x)
                feats = model(genomic_features=x, return_features=True) if not train else None
            if nll:
                hz, S, Yh, _ = res
                # train_loop_survival passes hazards [1 x B x K]; with B == 1 the loss sees [1,1,K];
                # use the 2-D view the loss indexing needs (len(Y) == 1)
                loss = lu.NLLSurvLoss(alpha=0.15)(hazards=hz[0], S=S[0], Y=torch.tensor([y]),
                                                  c=torch.tensor([float(c_np[0])]))
                loss.backward()
                _record(out, f"{name}/{tag}", hz, S, Yh, loss, model, None, feats)
            else:
                risk = res[0]
                loss = lu.CoxSurvLoss()(risks=risk, times=torch.tensor(t_np), c=torch.as_tensor(c_np).to(dt))
                loss.backward()
                _record(out, f"{name}/{tag}", risk.reshape(-1), None, None, loss, model, None, feats)
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "omic.npz"), **out)
    print("omic.npz", len(out), "entries")


MM_CASES = [
    # name, fusion, mode, N_path, n_radio, G, gate_path, gate_radio, K
    ("concat_rpo", "concat", "radio_path_omic", 600, 64, 80, True, True, 4),
    ("concat_po_u", "concat", "path_omic", 200, 0, 36, False, True, 4),
    ("tensor_rpo", "tensor", "radio_path_omic", 300, 48, 80, True, True, 4),
    ("tensor_rp", "tensor", "radio_path", 150, 32, 80, True, False, 8),
]


def gen_mm(mm, lu):
    out, meta = {}, {}
    mm.size_path = [1024, 256, 256]            # shim 1: undefined global (model_mm_attention_mil.py:83)
    orig_ft = torch.cuda.FloatTensor
    torch.cuda.FloatTensor = torch.FloatTensor  # shim 3: model_modules.py:164 hard-codes the CUDA ctor
    try:
        for (name, fusion, mode, Np, nr, G, gp, gr, K) in MM_CASES:
            seed = 700 + len(meta)
            sd_np = gen.mm_state_dict(seed=seed, input_dim=G, fusion=fusion, gate_path=gp, gate_radio=gr,
                                      dropout=False, n_classes=K, mode=mode, n_mod=4, bias_std=0.05)
            xs_np = [gen.bag(seed + 5000, max(nr, 1), stream=7 * i) for i in range(4)]
            xp_np = gen.bag(seed + 5000, max(Np, 1), stream=100)
            xo_np = gen.normal(seed + 5000, (G,), stream=200)
            y, c, alpha = 2, 0, 0.0
            meta[name] = dict(fusion=fusion, mode=mode, Np=Np, nr=nr, G=G, gate_path=gp, gate_radio=gr, K=K,
                              seed=seed, x_seed=seed + 5000, y=y, c=c, alpha=alpha, bias_std=0.05,
                              shims=["size_path", "base_init", "genomic_1d", "cuda.FloatTensor->cpu"])
            # fp32 only for 'tensor' (the FloatTensor shim fixes the dtype of the appended ones)
            dts = ((torch.float32, "f32"),) if fusion == "tensor" else ((torch.float32, "f32"), (torch.float64, "f64"))
            for dt, tag in dts:
                model = mm.MM_MIL_Attention_fc_surv.__new__(mm.MM_MIL_Attention_fc_surv)
                mm.MM_MIL_Attention_fc.__init__(                       # shim 2: base-class ctor
                    model, input_dim=G, radio_fusion="concat", fusion=fusion, gate=True, gate_path=gp,
                    gate_radio=gr, dropout=False, model_size_radio="small", model_size_wsi="small",
                    model_size_omic="small", n_classes=K, mode=mode)
                _load_sd(model, sd_np, dt)
                model.eval()
                kw = {m: torch.as_tensor(x).to(dt) for m, x in zip(MODS, xs_np)}
                kw["path_features"] = torch.as_tensor(xp_np).to(dt)
                kw["genomic_features"] = torch.as_tensor(xo_np).to(dt)
                hz, S, Yh, A_raw = model(**kw)
                loss = lu.NLLSurvLoss(alpha=alpha)(hazards=hz, S=S, Y=torch.tensor([y]), c=torch.tensor([float(c)]))
                loss.backward()
                _record(out, f"{name}/{tag}", hz, S, Yh, loss, model, A_raw, None)
    finally:
        torch.cuda.FloatTensor = orig_ft
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "mm.npz"), **out)
    print("mm.npz", len(out), "entries")


def gen_trajectory(mp, lu, uu):
    """Eval-mode-dropout re-drive of utils/core_utils.py:200-247 for 4 bags, gc=2, reg_type=all."""
    out = {}
    K, gc, lam, lr, wd = 4, 2, 1e-5, 2e-4, 1e-5
    seed = 900
    sd_np = gen.path_state_dict(seed=seed, gated=True, size="small", n_classes=K, bias_std=0.05)
    bags = [(gen.bag(seed + 5000 + i, n), y, c) for i, (n, y, c) in
            enumerate([(400, 1, 0), (250, 3, 1), (333, 0, 0), (128, 2, 1)])]
    meta = dict(K=K, gc=gc, lambda_reg=lam, lr=lr, reg=wd, seed=seed, alpha=0.0,
                bags=[dict(n=int(b[0].shape[0]), y=b[1], c=b[2], x_seed=seed + 5000 + i) for i, b in enumerate(bags)])
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        model = mp.MIL_Attention_fc_surv_path(gate_path=True, model_size_wsi="small", dropout=False, n_classes=K)
        _load_sd(model, sd_np, dt)
        model.eval()   # dropout disabled; everything else as train_loop_survival
        opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=wd)   # utils/utils.py:144-146
        loss_fn = lu.NLLSurvLoss(alpha=0.0)
        losses, risks = [], []
        for bi, (x_np, y, c) in enumerate(bags):
            hz, S, Yh, _ = model(path_features=torch.as_tensor(x_np).to(dt))
            risk = -torch.sum(S, dim=1)
            loss = loss_fn(hazards=hz, S=S, Y=torch.tensor([y]), c=torch.tensor([float(c)]))
            losses.append(loss.item())
            risks.append(risk.item())
            loss_reg = uu.l1_reg_all(model) * lam
            loss = loss / gc + loss_reg
            loss.backward()
            if (bi + 1) % gc == 0:
                opt.step()
                opt.zero_grad()
                for k, p in model.named_parameters():
                    summarize(f"{tag}/step{(bi + 1) // gc}/{k}", p.detach().double().numpy(), out)
        out[f"{tag}/losses"] = np.array(losses)
        out[f"{tag}/risks"] = np.array(risks)
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "trajectory.npz"), **out)
    print("trajectory.npz", len(out), "entries")


def gen_init(mp, mr, mg):
    """Same-seed construction: parameter summaries after torch.manual_seed(1) (main.py:47, 197-207)."""
    out = {}
    for name, ctor in (
        ("path_g", lambda: mp.MIL_Attention_fc_surv_path(gate_path=True, n_classes=4)),
        ("path_u_do", lambda: mp.MIL_Attention_fc_surv_path(gate_path=False, dropout=True, n_classes=8)),
        ("radio", lambda: mr.MIL_Attention_fc_surv_radio(n_classes=4)),
        ("maxnet", lambda: mg.MaxNet(input_dim=36, bag_loss="cox_surv")),
    ):
        torch.manual_seed(1)
        m = ctor()
        out[f"{name}/keys"] = np.array(json.dumps([(k, list(v.shape)) for k, v in m.state_dict().items()]))
        for k, v in m.state_dict().items():
            summarize(f"{name}/{k}", v.double().numpy(), out, full_below=64)
    np.savez_compressed(os.path.join(OUT, "init.npz"), **out)
    print("init.npz", len(out), "entries")


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    mp, mr, mg, mm, mods, lu, uu = _import_reference()
    gen_path(mp, lu)
    gen_radio(mr, lu)
    gen_omic(mg, lu)
    gen_mm(mm, lu)
    gen_trajectory(mp, lu, uu)
    gen_init(mp, mr, mg)


if __name__ == "__main__":
    main()
