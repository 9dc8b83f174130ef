"""GPU: K-split projections of short grids (LinearParams::ksplit; include/mmf_amil.h: mmf_amil_desc::sync).

A 1,000-instance bag is 64 projection tiles with a 32-chunk K loop each; the radio head's reduce_dim (M = 512, K = 4 x 1024)
128 tiles with 128 chunks.  With tick words the launcher splits K four ways: every workgroup writes a partial tile, the last
to arrive at a tile sums the partials in split order and runs the epilogue (bias, ReLU, dropout, relu bits).  Checked here:
the split really runs (kernel trace: grid size is not visible, so through the workspace-bytes entry point and the results'
independence of the tick words), results equal the fp64 oracle, are bit-reproducible run to run, equal the unsplit launch to
fp32 rounding, and the tick words are zero again after every call."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import cases
from oracle import inputs as gen
from test_gpu_path import DEV, _t

pytestmark = pytest.mark.gpu


def _linear(xs, W, b, act=0, drop_p=0.0, seed=0, site=0, split=True):
    from multimodalfusion_amd import _lib, ops
    l = _lib.lib()
    M, kseg = xs[0].shape
    N = W.shape[0]
    y = torch.empty((M, N), dtype=torch.float32, device=DEV)
    segs = (C.c_void_p * len(xs))(*[x.data_ptr() for x in xs])
    wsb = l.mmf_linear_forward_workspace_bytes(M, N, len(xs), kseg) if split else 0
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)
    sw = ops.sync_words(DEV)
    rc = l.mmf_linear_forward(segs, len(xs), kseg, M, C.c_void_p(W.data_ptr()), C.c_void_p(b.data_ptr()), N, act,
                              C.c_float(drop_p), seed, site, None, C.c_void_p(y.data_ptr()),
                              C.c_void_p(ws.data_ptr()) if wsb else None, wsb, C.c_void_p(sw.data_ptr()) if wsb else None,
                              ops.SYNC_WORDS if wsb else 0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert int(sw.abs().sum()) == 0                     # every call leaves the tick words zero
    return y, wsb


@pytest.mark.parametrize("M,nseg,kseg,N", [(512, 4, 1024, 1024), (500, 4, 1024, 1024), (1000, 1, 1024, 256), (77, 1, 1024, 256),
                                          (2048, 1, 1024, 256), (512, 2, 512, 256)])
def test_ksplit_linear_matches_fp64_and_the_unsplit_launch(M, nseg, kseg, N):
    xs = [gen.normal(31 + i, (M, kseg), stream=i) for i in range(nseg)]
    W = gen.normal(41, (N, nseg * kseg), stream=5, std=0.05)
    b = gen.normal(42, (N,), stream=6, std=0.3)
    txs, tW, tb = [_t(x) for x in xs], _t(W), _t(b)
    for act, drop_p in ((0, 0.0), (1, 0.25)):
        y, wsb = _linear(txs, tW, tb, act, drop_p, seed=9, site=2, split=True)
        assert wsb > 0                                  # these shapes ARE split
        y0, _ = _linear(txs, tW, tb, act, drop_p, seed=9, site=2, split=False)
        pre = np.concatenate(xs, axis=1).astype(np.float64) @ W.astype(np.float64).T + b.astype(np.float64)
        ref = np.maximum(pre, 0) if act == 1 else pre
        if drop_p > 0:
            ref = np.where(gen.keep_mask(9, 2, M, N, drop_p), ref / (1 - drop_p), 0.0)
        np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=1e-4, atol=3e-5)
        np.testing.assert_allclose(y.cpu().numpy(), y0.cpu().numpy(), rtol=1e-5, atol=2e-5)
        for _ in range(5):                              # bit-reproducible: partials are summed in split order, by whoever is last
            y2, _ = _linear(txs, tW, tb, act, drop_p, seed=9, site=2, split=True)
            assert torch.equal(y, y2)


def test_large_shapes_are_not_split():
    from multimodalfusion_amd import _lib
    l = _lib.lib()
    assert l.mmf_linear_forward_workspace_bytes(50000, 256, 1, 1024) == 0
    assert l.mmf_linear_forward_workspace_bytes(512, 1024, 4, 1024) > 0


@pytest.mark.parametrize("N", [1000, 1, 333])
def test_small_bag_step_with_and_without_tick_words(N, monkeypatch):
    """The path head's one-call step on a small bag: with the tick words (K-split projection) and without them (sync = NULL:
    one workgroup per tile) -- both against the fp64 oracle at the usual bars, bit-reproducible, words left zero."""
    from multimodalfusion_amd import ops
    from multimodalfusion_amd.models import MIL_Attention_fc_surv_path
    m = dict(N=N, gated=True, size="small", K=4, dropout=False, y=1, c=0, alpha=0.0, bias_std=0.05, train=True, seed=21,
             x_seed=22, mask_seed=5)
    sd, x, _ = cases.path_inputs(m)
    ref = cases.run_path(m)
    model = MIL_Attention_fc_surv_path(gate_path=True, n_classes=4)
    model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    model = model.to(DEV).train()
    monkeypatch.setattr(ops, "next_dropout_seed", lambda: m["mask_seed"])
    Y, c = torch.tensor([1], device=DEV), torch.tensor([0.0], device=DEV)
    xt = _t(x)
    outs = []
    for use_sync in (True, True, False):
        if not use_sync:
            monkeypatch.setattr(ops, "sync_words", lambda device=None: None)
        grads = [torch.zeros_like(p) for p in model.parameters()]
        out = model.nll_step(xt, Y, c, alpha=0.0, grad_out=grads, accumulate=False)
        torch.cuda.synchronize()
        outs.append((out[4].item(), out[3].cpu().numpy().copy(), [g.cpu().numpy().copy() for g in grads]))
    monkeypatch.undo()
    assert int(ops.sync_words(DEV).abs().sum()) == 0
    assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1])
    for a, b in zip(outs[0][2], outs[1][2]):
        assert np.array_equal(a, b)
    for loss, A, grads in (outs[0], outs[2]):
        assert abs(loss - float(ref["loss"])) < 1e-5
        np.testing.assert_allclose(A.reshape(-1), ref["A_raw"].reshape(-1), rtol=0, atol=1e-4)
        for (k, _), g in zip(model.named_parameters(), grads):
            r = ref["grads"][k]
            assert np.abs(g - r).max() <= 1e-5 + 1e-4 * np.abs(r).max(), k
