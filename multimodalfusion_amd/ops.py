"""torch.autograd Functions over the C ABI (include/mmf_amil.h).

PyTorch is used for device memory, streams and the autograd graph only; every arithmetic
step of the path runs in the HIP kernels of libmmf_amil.so.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib
from ._lib import AmilDesc, AmilGrads, NllTarget, SurvHead, check, lib, ptr, stream_ptr

ACT = {"none": 0, "relu": 1, "tanh": 2, "sigmoid": 3, "selu": 4}

_drop_calls = 0

# Host-side plumbing state for two optional per-call ABI arguments (the library itself keeps no state):
#   _seed_word  a device int32 word added to every dropout seed (graph.DeviceSeed sets it while it captures / replays a
#               hipGraph); each autograd node remembers the word its forward used, so its backward uses the same one;
#   _trace      an mmf_trace handle (bench.py's roofline leg: per-kernel HIP-event timing on the launch stream).
#   _concurrent scheduling hint (mmf_amil_desc::concurrent): pipeline.BagsInFlight raises it while it issues a bag.
#   _gemm       mmf_amil_desc::gemm: 0 = exact-fp32 MFMA, 1 = split-operand bf16x3 (fp32-equivalent error, faster on large
#               bags); starts from the MMF_GEMM environment variable (0 / 1, default 0).
_seed_word = None
_trace = None
_concurrent = 0
_gemm = int(os.environ.get("MMF_GEMM", "0"))


def set_gemm(mode):
    """mode: 0 (exact fp32 MFMA) or 1 (bf16x3 split operands).  Returns the previous one."""
    global _gemm
    prev, _gemm = _gemm, int(mode)
    return prev


def set_concurrent(flag):
    global _concurrent
    prev, _concurrent = _concurrent, 1 if flag else 0
    return prev


def set_device_seed(word):
    """word: int32 CUDA tensor [1] or None.  Returns the previous one."""
    global _seed_word
    prev, _seed_word = _seed_word, word
    return prev


def set_trace(handle):
    global _trace
    prev, _trace = _trace, handle
    return prev


SYNC_WORDS = 1024
_sync = {}
_sync_override = None
_have_gpu = None          # torch.cuda.is_available(), asked once (it reads the environment on every call)


def set_sync_override(words):
    """words: zeroed int32 CUDA tensor [SYNC_WORDS] that every call made from now on uses, or None.  Returns the previous one.
    graph.GraphedStep gives each captured graph its own (replays of two graphs may overlap on different streams)."""
    global _sync_override
    prev, _sync_override = _sync_override, words
    return prev


def sync_words(device=None):
    """The tick words of mmf_amil_desc::sync for calls issued on the CURRENT stream of `device`: one zeroed int32 tensor per
    (device, stream), made once and kept -- every call leaves the words zero, and calls that may overlap (other streams: bags in
    flight) get their own.  None while a stream is being captured without an override (the calls then run unsplit)."""
    if _sync_override is not None:
        return _sync_override
    global _have_gpu
    if device is not None and not isinstance(device, torch.device):
        device = torch.device(device)
    if device is not None and device.type != "cuda":
        return None                      # CPU tensors: the call itself raises MmfError (there is no CPU path)
    if _have_gpu is None:
        _have_gpu = torch.cuda.is_available()
    if not _have_gpu:
        return None
    if torch.cuda.is_current_stream_capturing():
        return None
    index = torch._C._cuda_getDevice() if device is None or device.index is None else device.index
    key = (index, torch._C._cuda_getCurrentRawStream(index))
    t = _sync.get(key)
    if t is None:
        t = _sync[key] = torch.zeros(SYNC_WORDS, dtype=torch.int32, device=torch.device("cuda", index))
    return t


def _amil_desc(N, L, H, D, gated, W1, b1, Wa, ba, Wb, bb, Wc, bc, p_h, p_att, seed, seed_word, concurrent=None):
    sw = sync_words(W1.device if W1 is not None else None)
    return AmilDesc(sync=ptr(sw), sync_words=SYNC_WORDS if sw is not None else 0,
                    N=N, L=L, H=H, D=D, gated=1 if gated else 0,
                    W1=ptr(W1), b1=ptr(b1), Wa=ptr(Wa), ba=ptr(ba),
                    Wb=ptr(Wb) if gated else None, bb=ptr(bb) if gated else None,
                    Wc=ptr(Wc), bc=ptr(bc), p_h=float(p_h), p_att=float(p_att), seed=int(seed) & 0xFFFFFFFF,
                    seed_dev=ptr(seed_word), trace=_trace, concurrent=_concurrent if concurrent is None else concurrent,
                    gemm=_gemm)


def next_dropout_seed() -> int:
    """Per-call dropout seed: deterministic given torch.manual_seed(), no device sync."""
    global _drop_calls
    _drop_calls += 1
    return (torch.initial_seed() * 0x9E3779B1 + _drop_calls * 0x85EBCA6B) & 0xFFFFFFFF


def _f32c(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise _lib.MmfError(f"expected float32 tensor, got {t.dtype}")
    return t.contiguous()


class AmilPoolFn(torch.autograd.Function):
    """(x, attention-stack params) -> (M [1 x H], A_raw [1 x N]).

    Mirrors `A, h = attention_net(x); A = A.T; A_raw = A; M = softmax(A) @ h`
    (models/model_attention_mil_path.py:52-56 in the reference).
    """

    @staticmethod
    def forward(ctx, x, W1, b1, Wa, ba, Wb, bb, Wc, bc, gated, p_h, p_att, seed, M_out=None):
        # (M_out: only for callers that run the node by hand -- the multimodal step lets the stack write its embedding
        # straight into its slot of the concatenated feature vector)
        # a bf16 bag selects the bf16-storage kernels (include/mmf_amil.h: mmf_amil_bf16_*); parameters stay fp32
        bf16 = x.dtype == torch.bfloat16
        x = x.contiguous() if bf16 else _f32c(x)
        W1, b1, Wa, ba, Wc, bc = map(_f32c, (W1, b1, Wa, ba, Wc, bc))
        Wb, bb = _f32c(Wb), _f32c(bb)
        if x.dim() != 2:
            raise _lib.MmfError(f"bag must be [N x L], got {tuple(x.shape)}")
        N, L = x.shape
        H, D = W1.shape[0], Wa.shape[0]
        if W1.shape[1] != L or Wa.shape[1] != H or Wc.numel() != D:
            raise _lib.MmfError("attention stack shapes do not match the bag")
        word = _seed_word
        d = _amil_desc(N, L, H, D, gated, W1, b1, Wa, ba, Wb, bb, Wc, bc, p_h, p_att, seed, word)
        l = lib()
        ws_fn, fwd_fn = ((l.mmf_amil_bf16_workspace_bytes, l.mmf_amil_bf16_forward) if bf16
                         else (l.mmf_amil_workspace_bytes, l.mmf_amil_forward))
        nbytes = ws_fn(N, L, H, D, d.gated)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        M = torch.empty((1, H), dtype=torch.float32, device=x.device) if M_out is None else M_out
        A_raw = torch.empty((1, N), dtype=torch.float32, device=x.device)
        check(fwd_fn(C.byref(d), ptr(x), ptr(ws), nbytes, ptr(M), ptr(A_raw), stream_ptr()),
              "mmf_amil_bf16_forward" if bf16 else "mmf_amil_forward")
        ctx.bf16 = bf16
        ctx.desc_args = (N, L, H, D, bool(gated), float(p_h), float(p_att), int(seed) & 0xFFFFFFFF)
        ctx.seed_word = word
        ctx.concurrent = _concurrent          # the backward runs on the autograd thread, after the caller has lowered the hint
        ctx.ws = ws
        ctx.save_for_backward(x, W1, b1, Wa, ba, Wb, bb, Wc, bc, M, A_raw)
        ctx.set_materialize_grads(False)      # an unused output (A_raw, mostly) must not cost a zero-fill launch
        return M, A_raw

    @staticmethod
    def backward(ctx, gM, gA):
        x, W1, b1, Wa, ba, Wb, bb, Wc, bc, M, A_raw = ctx.saved_tensors
        N, L, H, D, gated, p_h, p_att, seed = ctx.desc_args
        dev = x.device
        gM = torch.zeros((1, H), dtype=torch.float32, device=dev) if gM is None else _f32c(gM)
        gA = _f32c(gA) if gA is not None else None
        d = _amil_desc(N, L, H, D, gated, W1, b1, Wa, ba, Wb, bb, Wc, bc, p_h, p_att, seed, ctx.seed_word, ctx.concurrent)
        new = lambda ref: torch.empty_like(ref)
        dW1, db1, dWa, dba, dWc, dbc = new(W1), new(b1), new(Wa), new(ba), new(Wc), new(bc)
        dWb, dbb = (new(Wb), new(bb)) if gated else (None, None)
        if ctx.bf16 and ctx.needs_input_grad[0]:
            raise _lib.MmfError("a bf16 bag is a leaf: no input gradient on the bf16 path")
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        g = AmilGrads(dW1=ptr(dW1), db1=ptr(db1), dWa=ptr(dWa), dba=ptr(dba), dWb=ptr(dWb), dbb=ptr(dbb),
                      dWc=ptr(dWc), dbc=ptr(dbc), dx=ptr(dx))
        ws = ctx.ws
        bwd_fn = lib().mmf_amil_bf16_backward if ctx.bf16 else lib().mmf_amil_backward
        check(bwd_fn(C.byref(d), ptr(x), ptr(ws), ws.numel(), ptr(M), ptr(A_raw),
                     ptr(gM), ptr(gA), C.byref(g), stream_ptr()), "mmf_amil_bf16_backward" if ctx.bf16 else "mmf_amil_backward")
        return dx, dW1, db1, dWa, dba, dWb, dbb, dWc, dbc, None, None, None, None


def amil_infer(x, W1, b1, Wa, ba, Wb, bb, Wc, bc, gated):
    """Forward-only attention stack (include/mmf_amil.h: mmf_amil[_bf16]_infer): nothing is saved for a backward."""
    bf16 = x.dtype == torch.bfloat16
    x = x.contiguous() if bf16 else _f32c(x)
    W1, b1, Wa, ba, Wc, bc = map(_f32c, (W1, b1, Wa, ba, Wc, bc))
    Wb, bb = _f32c(Wb), _f32c(bb)
    if x.dim() != 2:
        raise _lib.MmfError(f"bag must be [N x L], got {tuple(x.shape)}")
    N, L = x.shape
    H, D = W1.shape[0], Wa.shape[0]
    if W1.shape[1] != L or Wa.shape[1] != H or Wc.numel() != D:
        raise _lib.MmfError("attention stack shapes do not match the bag")
    d = _amil_desc(N, L, H, D, gated, W1, b1, Wa, ba, Wb, bb, Wc, bc, 0.0, 0.0, 0, None)
    l = lib()
    ws_fn, fn, name = ((l.mmf_amil_bf16_infer_workspace_bytes, l.mmf_amil_bf16_infer, "mmf_amil_bf16_infer") if bf16
                       else (l.mmf_amil_infer_workspace_bytes, l.mmf_amil_infer, "mmf_amil_infer"))
    nbytes = ws_fn(N, L, H, D, d.gated)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    M = torch.empty((1, H), dtype=torch.float32, device=x.device)
    A_raw = torch.empty((1, N), dtype=torch.float32, device=x.device)
    check(fn(C.byref(d), ptr(x), ptr(ws), nbytes, ptr(M), ptr(A_raw), stream_ptr()), name)
    return M, A_raw


def amil_pool(x, W1, b1, Wa, ba, Wb, bb, Wc, bc, gated, p_h=0.0, p_att=0.0, seed=0):
    if not torch.is_grad_enabled() and p_h == 0.0 and p_att == 0.0:
        return amil_infer(x, W1, b1, Wa, ba, Wb, bb, Wc, bc, gated)     # torch.no_grad() + eval: the inference consumers
    return AmilPoolFn.apply(x, W1, b1, Wa, ba, Wb, bb, Wc, bc, gated, p_h, p_att, seed)


class AmilHeadFn(torch.autograd.Function):
    """Attention stack + classifier/hazard head as ONE autograd node (models/model_attention_mil_path.py:52-61):
    (x, stack params, Wk, bk) -> (hazards, S, Y_hat, A_raw).  Same kernels as AmilPoolFn + SurvHeadFn; one node less
    on a path where a 1k-10k bag step is bound by host dispatch (tools/host_split2.py)."""

    @staticmethod
    def forward(ctx, x, W1, b1, Wa, ba, Wb, bb, Wc, bc, Wk, bk, gated, p_h, p_att, seed):
        bf16 = x.dtype == torch.bfloat16
        x = x.contiguous() if bf16 else _f32c(x)
        W1, b1, Wa, ba, Wc, bc, Wk, bk = map(_f32c, (W1, b1, Wa, ba, Wc, bc, Wk, bk))   # model.half() / .double() must not
        Wb, bb = _f32c(Wb), _f32c(bb)                                                    # reach the fp32 kernels
        if x.dim() != 2:
            raise _lib.MmfError(f"bag must be [N x L], got {tuple(x.shape)}")
        N, L = x.shape
        H, D, K = W1.shape[0], Wa.shape[0], Wk.shape[0]
        if W1.shape[1] != L or Wa.shape[1] != H or Wc.numel() != D or Wk.shape[1] != H:
            raise _lib.MmfError("attention stack / classifier shapes do not match the bag")
        seed = int(seed) & 0xFFFFFFFF
        word = _seed_word
        d = _amil_desc(N, L, H, D, gated, W1, b1, Wa, ba, Wb, bb, Wc, bc, p_h, p_att, seed, word)
        l = lib()
        nbytes = (l.mmf_amil_bf16_workspace_bytes if bf16 else l.mmf_amil_workspace_bytes)(N, L, H, D, d.gated)
        dev = x.device
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        M = torch.empty((1, H), dtype=torch.float32, device=dev)
        A_raw = torch.empty((1, N), dtype=torch.float32, device=dev)
        out = torch.empty((3, 1, K), dtype=torch.float32, device=dev)        # logits, hazards, S
        Y_hat = torch.empty((1, 1), dtype=torch.int64, device=dev)
        st = stream_ptr()
        if K <= 32:      # the head runs as the tail of the pooling merge kernel: one launch less
            hd = SurvHead(Wk=ptr(Wk), bk=ptr(bk), K=K, logits=ptr(out[0]), hazards=ptr(out[1]), S=ptr(out[2]),
                          Y_hat=ptr(Y_hat), risk=None)
            check(l.mmf_amil_head_forward(C.byref(d), ptr(x), 1 if bf16 else 0, ptr(ws), nbytes, C.byref(hd), ptr(M),
                                          ptr(A_raw), st), "mmf_amil_head_forward")
        else:
            fwd_fn = l.mmf_amil_bf16_forward if bf16 else l.mmf_amil_forward
            check(fwd_fn(C.byref(d), ptr(x), ptr(ws), nbytes, ptr(M), ptr(A_raw), st), "mmf_amil_forward")
            check(l.mmf_surv_head_forward(ptr(M), ptr(Wk), ptr(bk), 1, H, K, ptr(out[0]), ptr(out[1]), ptr(out[2]),
                                          ptr(Y_hat), st), "mmf_surv_head_forward")
        ctx.cfg = (N, L, H, D, K, bool(gated), float(p_h), float(p_att), seed, bf16)
        ctx.seed_word = word
        ctx.ws = ws
        ctx.save_for_backward(x, W1, b1, Wa, ba, Wb, bb, Wc, bc, Wk, M, A_raw, out)
        ctx.mark_non_differentiable(Y_hat)
        ctx.set_materialize_grads(False)      # no zero-fill launches for the outputs the loss does not use (A_raw, Y_hat)
        return out[1], out[2], Y_hat, A_raw

    @staticmethod
    def backward(ctx, gH, gS, _gY, gA):
        x, W1, b1, Wa, ba, Wb, bb, Wc, bc, Wk, M, A_raw, out = ctx.saved_tensors
        N, L, H, D, K, gated, p_h, p_att, seed, bf16 = ctx.cfg
        if bf16 and ctx.needs_input_grad[0]:
            raise _lib.MmfError("a bf16 bag is a leaf: no input gradient on the bf16 path")
        dev = x.device
        l = lib()
        st = stream_ptr()
        gH = _f32c(gH) if gH is not None else None
        gS = _f32c(gS) if gS is not None else None
        gA = _f32c(gA) if gA is not None else None
        dM = torch.empty((1, H), dtype=torch.float32, device=dev)
        dWk = torch.empty_like(Wk)
        dbk = torch.empty((K,), dtype=torch.float32, device=dev)
        check(l.mmf_surv_head_backward(ptr(gH), ptr(gS), ptr(out[1]), ptr(M), ptr(Wk), 1, H, K,
                                       ptr(dM), ptr(dWk), ptr(dbk), st), "mmf_surv_head_backward")
        d = _amil_desc(N, L, H, D, gated, W1, b1, Wa, ba, Wb, bb, Wc, bc, p_h, p_att, seed, ctx.seed_word)
        new = torch.empty_like
        dW1, db1, dWa, dba, dWc, dbc = new(W1), new(b1), new(Wa), new(ba), new(Wc), new(bc)
        dWb, dbb = (new(Wb), new(bb)) if gated else (None, None)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        g = AmilGrads(dW1=ptr(dW1), db1=ptr(db1), dWa=ptr(dWa), dba=ptr(dba), dWb=ptr(dWb), dbb=ptr(dbb),
                      dWc=ptr(dWc), dbc=ptr(dbc), dx=ptr(dx))
        ws = ctx.ws
        bwd_fn = l.mmf_amil_bf16_backward if bf16 else l.mmf_amil_backward
        check(bwd_fn(C.byref(d), ptr(x), ptr(ws), ws.numel(), ptr(M), ptr(A_raw), ptr(dM), ptr(gA), C.byref(g), st),
              "mmf_amil_backward")
        return dx, dW1, db1, dWa, dba, dWb, dbb, dWc, dbc, dWk, dbk, None, None, None, None


def amil_nll_step(x, stack, Wk, bk, gated, Y, c, alpha, grads, loss_scale=1.0, accumulate=False, p_h=0.0, p_att=0.0,
                  seed=0, eps=1e-7, dx=None):
    """(dx: optional [N x L] fp32 tensor that receives d(loss * loss_scale)/dx -- the radio head, whose bag is
    reduce_dim's output.)
    One bag's whole training step in ONE C-ABI call (include/mmf_amil.h: mmf_amil_nll_step): attention stack +
    classifier / hazard head + nll_surv + backward.  No autograd graph is built.

    stack = (W1, b1, Wa, ba, Wb, bb, Wc, bc); grads = the matching gradient tensors (dW1, db1, dWa, dba, dWb, dbb, dWc,
    dbc, dWk, dbk), written with d(loss * loss_scale) -- added to when `accumulate`.  Y, c: device tensors [1].
    Returns (hazards [1 x K], S [1 x K], Y_hat [1 x 1], A_raw [1 x N], loss (0-dim, unscaled), risk [1])."""
    bf16 = x.dtype == torch.bfloat16
    x = x.contiguous() if bf16 else _f32c(x)
    W1, b1, Wa, ba, Wb, bb, Wc, bc = stack
    W1, b1, Wa, ba, Wc, bc, Wk, bk = map(_f32c, (W1, b1, Wa, ba, Wc, bc, Wk, bk))
    Wb, bb = _f32c(Wb), _f32c(bb)
    if x.dim() != 2:
        raise _lib.MmfError(f"bag must be [N x L], got {tuple(x.shape)}")
    N, L = x.shape
    H, D, K = W1.shape[0], Wa.shape[0], Wk.shape[0]
    if W1.shape[1] != L or Wa.shape[1] != H or Wc.numel() != D or Wk.shape[1] != H or K > 32:
        raise _lib.MmfError("attention stack / classifier shapes do not match the bag")
    dW1, db1, dWa, dba, dWb, dbb, dWc, dbc, dWk, dbk = grads
    for g_, w_ in ((dW1, W1), (db1, b1), (dWa, Wa), (dba, ba), (dWc, Wc), (dbc, bc), (dWk, Wk), (dbk, bk)) + \
            (((dWb, Wb), (dbb, bb)) if gated else ()):
        if g_ is None or g_.dtype != torch.float32 or g_.shape != w_.shape or not g_.is_contiguous():
            raise _lib.MmfError("gradient buffers must be contiguous float32 tensors shaped like their parameters")
    dev = x.device
    if not Y.is_cuda and bool(((Y < 0) | (Y >= K)).any()):
        raise IndexError(f"nll_surv: label out of range [0, {K})")
    Y = Y.reshape(1).to(device=dev, dtype=torch.int64)
    c = c.reshape(1).to(device=dev, dtype=torch.float32)
    d = _amil_desc(N, L, H, D, gated, W1, b1, Wa, ba, Wb, bb, Wc, bc, p_h, p_att, seed, _seed_word)
    l = lib()
    nbytes = (l.mmf_amil_bf16_workspace_bytes if bf16 else l.mmf_amil_workspace_bytes)(N, L, H, D, d.gated)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    A_raw = torch.empty((1, N), dtype=torch.float32, device=dev)
    out = torch.empty((3 * K + 2,), dtype=torch.float32, device=dev)       # logits, hazards, S, loss, risk
    Y_hat = torch.empty((1, 1), dtype=torch.int64, device=dev)
    hd = SurvHead(Wk=ptr(Wk), bk=ptr(bk), K=K, logits=ptr(out[0:K]), hazards=ptr(out[K:2 * K]), S=ptr(out[2 * K:3 * K]),
                  Y_hat=ptr(Y_hat), risk=ptr(out[3 * K + 1:]))
    tg = NllTarget(Y=ptr(Y), c=ptr(c), alpha=float(alpha), eps=float(eps), loss_scale=float(loss_scale),
                   loss=ptr(out[3 * K:]), dWk=ptr(dWk), dbk=ptr(dbk), accumulate=1 if accumulate else 0)
    g = AmilGrads(dW1=ptr(dW1), db1=ptr(db1), dWa=ptr(dWa), dba=ptr(dba), dWb=ptr(dWb) if gated else None,
                  dbb=ptr(dbb) if gated else None, dWc=ptr(dWc), dbc=ptr(dbc), dx=ptr(dx))
    if dx is not None and (bf16 or dx.dtype != torch.float32 or dx.shape != x.shape or not dx.is_contiguous()):
        raise _lib.MmfError("dx must be a contiguous float32 tensor shaped like an fp32 bag")
    check(l.mmf_amil_nll_step(C.byref(d), ptr(x), 1 if bf16 else 0, ptr(ws), nbytes, C.byref(hd), C.byref(tg),
                              ptr(A_raw), C.byref(g), stream_ptr()), "mmf_amil_nll_step")
    return (out[K:2 * K].view(1, K), out[2 * K:3 * K].view(1, K), Y_hat, A_raw, out[3 * K].view(()), out[3 * K + 1:].view(1))


def amil_head(x, W1, b1, Wa, ba, Wb, bb, Wc, bc, Wk, bk, gated, p_h=0.0, p_att=0.0, seed=0):
    if not torch.is_grad_enabled() and p_h == 0.0 and p_att == 0.0:       # inference consumers: no-save kernels
        M, A_raw = amil_infer(x, W1, b1, Wa, ba, Wb, bb, Wc, bc, gated)
        hz, S, Y_hat = surv_head(M, Wk, bk)
        return hz, S, Y_hat, A_raw
    return AmilHeadFn.apply(x, W1, b1, Wa, ba, Wb, bb, Wc, bc, Wk, bk, gated, p_h, p_att, seed)


class AttnNetFn(torch.autograd.Function):
    """The attention scorer alone: x [N x L] -> A [N x 1] (models/model_modules.py:84-85 / :105-110)."""

    @staticmethod
    def forward(ctx, x, Wa, ba, Wb, bb, Wc, bc, gated, p_att, seed):
        x, Wa, ba, Wc, bc = map(_f32c, (x, Wa, ba, Wc, bc))
        Wb, bb = _f32c(Wb), _f32c(bb)
        if x.dim() != 2 or Wa.shape[1] != x.shape[1] or Wc.shape[0] != 1 or Wc.shape[1] != Wa.shape[0]:
            raise _lib.MmfError("Attn_Net: x must be [N x L] and the scorer must have n_classes = 1")
        N, H = x.shape
        D = Wa.shape[0]
        word = _seed_word
        d = AmilDesc(N=N, L=H, H=H, D=D, gated=1 if gated else 0, W1=None, b1=None, Wa=ptr(Wa), ba=ptr(ba),
                     Wb=ptr(Wb) if gated else None, bb=ptr(bb) if gated else None, Wc=ptr(Wc), bc=ptr(bc),
                     p_h=0.0, p_att=float(p_att), seed=int(seed) & 0xFFFFFFFF, seed_dev=ptr(word), trace=_trace)
        l = lib()
        nbytes = l.mmf_attn_net_workspace_bytes(N, H, D, d.gated)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        A = torch.empty((N, 1), dtype=torch.float32, device=x.device)
        check(l.mmf_attn_net_forward(C.byref(d), ptr(x), ptr(ws), nbytes, ptr(A), stream_ptr()), "mmf_attn_net_forward")
        ctx.cfg = (N, H, D, bool(gated), float(p_att), int(seed) & 0xFFFFFFFF)
        ctx.seed_word = word
        ctx.ws = ws
        ctx.save_for_backward(x, Wa, ba, Wb, bb, Wc, bc)
        return A

    @staticmethod
    def backward(ctx, gA):
        x, Wa, ba, Wb, bb, Wc, bc = ctx.saved_tensors
        N, H, D, gated, p_att, seed = ctx.cfg
        gA = _f32c(gA).reshape(N)
        d = AmilDesc(N=N, L=H, H=H, D=D, gated=1 if gated else 0, W1=None, b1=None, Wa=ptr(Wa), ba=ptr(ba),
                     Wb=ptr(Wb) if gated else None, bb=ptr(bb) if gated else None, Wc=ptr(Wc), bc=ptr(bc),
                     p_h=0.0, p_att=p_att, seed=seed, seed_dev=ptr(ctx.seed_word), trace=_trace)
        new = torch.empty_like
        dWa, dba, dWc, dbc = new(Wa), new(ba), new(Wc), new(bc)
        dWb, dbb = (new(Wb), new(bb)) if gated else (None, None)
        dx = new(x) if ctx.needs_input_grad[0] else None
        g = AmilGrads(dW1=None, db1=None, dWa=ptr(dWa), dba=ptr(dba), dWb=ptr(dWb), dbb=ptr(dbb), dWc=ptr(dWc),
                      dbc=ptr(dbc), dx=ptr(dx))
        ws = ctx.ws
        check(lib().mmf_attn_net_backward(C.byref(d), ptr(x), ptr(ws), ws.numel(), ptr(gA), C.byref(g), stream_ptr()),
              "mmf_attn_net_backward")
        return dx, dWa, dba, dWb, dbb, dWc, dbc, None, None, None


def attn_net(x, Wa, ba, Wb, bb, Wc, bc, gated, p_att=0.0, seed=0):
    return AttnNetFn.apply(x, Wa, ba, Wb, bb, Wc, bc, gated, p_att, seed)


class LinearCatFn(torch.autograd.Function):
    """y = cat(xs, dim=1) @ W.T + b without materialising the concatenation
    (models/model_attention_mil_radio.py:80-82)."""

    @staticmethod
    def forward(ctx, W, b, *xs):
        xs = [_f32c(x) for x in xs]
        W, b = _f32c(W), _f32c(b)
        M, kseg = xs[0].shape
        for x in xs:
            if tuple(x.shape) != (M, kseg):
                raise _lib.MmfError("all concatenated segments must have the same [M x k] shape")
        nseg = len(xs)
        N = W.shape[0]
        if W.shape[1] != nseg * kseg:
            raise _lib.MmfError("weight does not match the concatenated width")
        y = torch.empty((M, N), dtype=torch.float32, device=W.device)
        segs = (C.c_void_p * nseg)(*[ptr(x) for x in xs])
        wsb = lib().mmf_linear_forward_workspace_bytes(M, N, nseg, kseg)
        sw = sync_words(W.device) if wsb else None
        ws = torch.empty(wsb, dtype=torch.uint8, device=W.device) if sw is not None else None
        check(lib().mmf_linear_forward(segs, nseg, kseg, M, ptr(W), ptr(b), N, ACT["none"], 0.0, 0, 0, None,
                                       ptr(y), ptr(ws), wsb if ws is not None else 0, ptr(sw), SYNC_WORDS if sw is not None else 0,
                                       stream_ptr()), "mmf_linear_forward")
        ctx.save_for_backward(W, *xs)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        W, *xs = ctx.saved_tensors
        gy = _f32c(gy)
        M, kseg = xs[0].shape
        nseg = len(xs)
        N, K = W.shape
        l = lib()
        dW = torch.empty_like(W)
        db = torch.empty((N,), dtype=torch.float32, device=W.device) if ctx.has_bias else None
        need_dx = any(ctx.needs_input_grad[2:])
        if need_dx and nseg != 1:
            raise _lib.MmfError("input gradient of a concatenated linear is not provided (bags are leaves)")
        dx = torch.empty_like(xs[0]) if need_dx else None
        nbytes = l.mmf_linear_backward_workspace_bytes(M, N, K)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=W.device)
        segs = (C.c_void_p * nseg)(*[ptr(x) for x in xs])
        check(l.mmf_linear_backward(ptr(gy), segs, nseg, kseg, M, ptr(W), N, ptr(dW), ptr(db), ptr(dx),
                                    ptr(ws), nbytes, stream_ptr()), "mmf_linear_backward")
        return (dW, db) + ((dx,) if nseg == 1 else (None,) * nseg)


def linear_cat(xs, W, b):
    return LinearCatFn.apply(W, b, *xs)


class SurvHeadFn(torch.autograd.Function):
    """feat [B x F] -> hazards, S [B x K], Y_hat [B x 1] (models/model_attention_mil_path.py:58-61)."""

    @staticmethod
    def forward(ctx, feat, Wk, bk):
        feat, Wk, bk = _f32c(feat), _f32c(Wk), _f32c(bk)
        B, F = feat.shape
        K = Wk.shape[0]
        dev = feat.device
        logits = torch.empty((B, K), dtype=torch.float32, device=dev)
        hazards = torch.empty_like(logits)
        S = torch.empty_like(logits)
        Y_hat = torch.empty((B, 1), dtype=torch.int64, device=dev)
        check(lib().mmf_surv_head_forward(ptr(feat), ptr(Wk), ptr(bk), B, F, K, ptr(logits), ptr(hazards),
                                          ptr(S), ptr(Y_hat), stream_ptr()), "mmf_surv_head_forward")
        ctx.save_for_backward(feat, Wk, hazards)
        ctx.mark_non_differentiable(Y_hat)
        ctx.set_materialize_grads(False)
        return hazards, S, Y_hat

    @staticmethod
    def backward(ctx, gH, gS, _gY):
        feat, Wk, hazards = ctx.saved_tensors
        B, F = feat.shape
        K = Wk.shape[0]
        gH = _f32c(gH) if gH is not None else None
        gS = _f32c(gS) if gS is not None else None
        dfeat = torch.empty_like(feat)
        dWk = torch.empty_like(Wk)
        dbk = torch.empty((K,), dtype=torch.float32, device=feat.device)
        check(lib().mmf_surv_head_backward(ptr(gH), ptr(gS), ptr(hazards), ptr(feat), ptr(Wk), B, F, K,
                                           ptr(dfeat), ptr(dWk), ptr(dbk), stream_ptr()), "mmf_surv_head_backward")
        return dfeat, dWk, dbk


def surv_head(feat, Wk, bk):
    return SurvHeadFn.apply(feat, Wk, bk)


class NllSurvFn(torch.autograd.Function):
    """utils/loss_utils.py:22-39; loss and both input gradients come out of one launch."""

    @staticmethod
    def forward(ctx, hazards, S, Y, c, alpha, eps):
        hazards, S = _f32c(hazards), _f32c(S)
        B, K = hazards.shape
        if not Y.is_cuda:       # labels still on the host: validate for free (the reference's gather raises IndexError);
            if bool(((Y < 0) | (Y >= K)).any()):       # device labels are checked by the kernel (NaN loss, no stray access)
                raise IndexError(f"nll_surv: label out of range [0, {K})")
        Y = Y.reshape(B).to(device=hazards.device, dtype=torch.int64).contiguous()
        c = c.reshape(B).to(device=hazards.device, dtype=torch.float32).contiguous()
        loss = torch.empty((), dtype=torch.float32, device=hazards.device)
        g = torch.empty((2, B, K), dtype=torch.float32, device=hazards.device)     # [d/d hazards ; d/d S]
        check(lib().mmf_nll_surv(ptr(hazards), ptr(S), ptr(Y), ptr(c), B, K, float(alpha), float(eps),
                                 ptr(loss), ptr(g[0]), ptr(g[1]), stream_ptr()), "mmf_nll_surv")
        ctx.save_for_backward(g)
        return loss

    @staticmethod
    def backward(ctx, gl):
        (g,) = ctx.saved_tensors
        out = g * gl                      # one launch for both
        return out[0], out[1], None, None, None, None


def nll_surv(hazards, S, Y, c, alpha=0.4, eps=1e-7):
    return NllSurvFn.apply(hazards, S, Y, c, alpha, eps)


def surv_head_nll_step(feat, Wk, bk, Y, c, alpha, dWk, dbk, loss_scale=1.0, accumulate=False, eps=1e-7):
    """Classifier + hazard head + NLLSurvLoss(alpha) + their backward on a feature vector feat [1 x F] (F <= 1024) in ONE
    launch (mmf_surv_head_nll_step; models/model_mm_attention_mil.py:190-191 + utils/loss_utils.py:22-39): dWk / dbk get
    the gradient of loss * loss_scale (added when `accumulate`).
    Returns (hazards [1 x K], S [1 x K], Y_hat [1 x 1], loss (0-dim, unscaled), risk [1], dfeat [1 x F]), detached."""
    feat, Wk, bk = _f32c(feat), _f32c(Wk), _f32c(bk)
    F = feat.numel()
    K = Wk.shape[0]
    if Wk.shape[1] != F or F > 1024 or K > 32:
        raise _lib.MmfError("surv_head_nll_step: classifier does not match the feature vector (F <= 1024, K <= 32)")
    for g_, w_ in ((dWk, Wk), (dbk, bk)):
        if g_ is None or g_.dtype != torch.float32 or g_.shape != w_.shape or not g_.is_contiguous():
            raise _lib.MmfError("gradient buffers must be contiguous float32 tensors shaped like their parameters")
    dev = feat.device
    if not Y.is_cuda and bool(((Y < 0) | (Y >= K)).any()):
        raise IndexError(f"nll_surv: label out of range [0, {K})")
    Y = Y.reshape(1).to(device=dev, dtype=torch.int64)
    c = c.reshape(1).to(device=dev, dtype=torch.float32)
    out = torch.empty((3 * K + 2,), dtype=torch.float32, device=dev)       # logits, hazards, S, loss, risk
    Y_hat = torch.empty((1, 1), dtype=torch.int64, device=dev)
    dfeat = torch.empty((1, F), dtype=torch.float32, device=dev)
    hd = SurvHead(Wk=ptr(Wk), bk=ptr(bk), K=K, logits=ptr(out[0:K]), hazards=ptr(out[K:2 * K]), S=ptr(out[2 * K:3 * K]),
                  Y_hat=ptr(Y_hat), risk=ptr(out[3 * K + 1:]))
    tg = NllTarget(Y=ptr(Y), c=ptr(c), alpha=float(alpha), eps=float(eps), loss_scale=float(loss_scale),
                   loss=ptr(out[3 * K:]), dWk=ptr(dWk), dbk=ptr(dbk), accumulate=1 if accumulate else 0)
    check(lib().mmf_surv_head_nll_step(ptr(feat), F, C.byref(hd), C.byref(tg), ptr(dfeat), stream_ptr()),
          "mmf_surv_head_nll_step")
    return (out[K:2 * K].view(1, K), out[2 * K:3 * K].view(1, K), Y_hat, out[3 * K].view(()), out[3 * K + 1:].view(1), dfeat)


class HandCtx:
    """Stands in for the autograd context when a node's forward / backward are run by hand (no graph): the multimodal
    one-call step drives AmilPoolFn / LinearCatFn directly, on the streams it chooses."""

    def __init__(self, needs_input_grad):
        self.needs_input_grad = tuple(needs_input_grad)
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def set_materialize_grads(self, value):
        pass

    def mark_non_differentiable(self, *tensors):
        pass


class CoxSurvFn(torch.autograd.Function):
    """utils/loss_utils.py:124-139."""

    @staticmethod
    def forward(ctx, risks, times, c):
        shape = risks.shape
        r = _f32c(risks.reshape(-1))
        B = r.numel()
        t = times.reshape(B).to(device=r.device, dtype=torch.float64).contiguous()
        cc = c.reshape(B).to(device=r.device, dtype=torch.float32).contiguous()
        loss = torch.empty((), dtype=torch.float32, device=r.device)
        dr = torch.empty_like(r)
        check(lib().mmf_cox_surv(ptr(r), ptr(t), ptr(cc), B, ptr(loss), ptr(dr), stream_ptr()), "mmf_cox_surv")
        ctx.save_for_backward(dr)
        ctx.shape = shape
        return loss

    @staticmethod
    def backward(ctx, g):
        (dr,) = ctx.saved_tensors
        return (dr * g).reshape(ctx.shape), None, None


def cox_surv(risks, times, c):
    return CoxSurvFn.apply(risks, times, c)


def maxnet_cox_step(x, W0, b0, W1, b1, Wc, bc, times, c, grads, loss_scale=1.0, accumulate=False, p_drop=0.0, seed=0):
    """MaxNet forward + CoxSurvLoss + backward of one batch in ONE launch (mmf_maxnet_cox_step; models/model_genomic.py:
    53-72 + utils/loss_utils.py:124-139).  times: float64 CUDA tensor [B]; grads: (dW0, db0, dW1, db1, dWc, dbc) tensors the
    gradients of loss * loss_scale are written to (or added to, `accumulate`).  Returns (risk [B], loss), detached."""
    x = _f32c(x)
    B, G = x.shape
    dev = x.device
    if times.dtype != torch.float64 or not times.is_cuda:
        raise _lib.MmfError("maxnet_cox_step: event times must be a float64 CUDA tensor")
    sw = sync_words(dev)
    if sw is None:
        raise _lib.MmfError("maxnet_cox_step needs tick words (ops.set_sync_override inside a stream capture)")
    cc = c.reshape(B).to(device=dev, dtype=torch.float32).contiguous()
    tt = times.reshape(B).contiguous()
    d = _lib.MaxnetDesc(B=B, G=G, H0=W0.shape[0], H1=W1.shape[0], x=ptr(x), W0=ptr(W0), b0=ptr(b0), W1=ptr(W1), b1=ptr(b1),
                        Wc=ptr(Wc), bc=ptr(bc), p_drop=float(p_drop), seed=int(seed) & 0xFFFFFFFF, seed_dev=ptr(_seed_word),
                        sync=ptr(sw), sync_words=SYNC_WORDS, trace=_trace)
    l = lib()
    nbytes = l.mmf_maxnet_cox_step_workspace_bytes(B)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    risk = torch.empty((B,), dtype=torch.float32, device=dev)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    g = _lib.MaxnetGrads(*[ptr(t) for t in grads])
    check(l.mmf_maxnet_cox_step(C.byref(d), ptr(tt), ptr(cc), float(loss_scale), ptr(ws), nbytes, ptr(risk), ptr(loss),
                                C.byref(g), 1 if accumulate else 0, stream_ptr()), "mmf_maxnet_cox_step")
    return risk, loss


DROP_KIND = {"none": 0, "dropout": 1, "alpha": 2}


class DenseFn(torch.autograd.Function):
    """y = drop(act(x @ W.T + b)) for small / odd-shaped layers (SNN blocks, fusion MLPs, classifiers)."""

    @staticmethod
    def forward(ctx, x, W, b, act, drop_kind, drop_p, seed, site):
        x, W, b = _f32c(x), _f32c(W), _f32c(b)
        B, K = x.shape
        N = W.shape[0]
        if W.shape[1] != K:
            raise _lib.MmfError(f"dense: weight {tuple(W.shape)} does not match input {tuple(x.shape)}")
        y = torch.empty((B, N), dtype=torch.float32, device=x.device)
        word = _seed_word
        check(lib().mmf_dense_forward(ptr(x), ptr(W), ptr(b), B, K, N, ACT[act], DROP_KIND[drop_kind], float(drop_p),
                                      int(seed) & 0xFFFFFFFF, int(site), ptr(word), ptr(y), stream_ptr()), "mmf_dense_forward")
        ctx.cfg = (act, drop_kind, float(drop_p), int(seed) & 0xFFFFFFFF, int(site), b is not None)
        ctx.seed_word = word
        ctx.save_for_backward(x, W, y)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, W, y = ctx.saved_tensors
        act, drop_kind, drop_p, seed, site, has_bias = ctx.cfg
        gy = _f32c(gy)
        B, K = x.shape
        N = W.shape[0]
        dpre = torch.empty_like(y)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dW = torch.empty_like(W)
        db = torch.empty((N,), dtype=torch.float32, device=x.device) if has_bias else None
        check(lib().mmf_dense_backward(ptr(gy), ptr(y), ptr(x), ptr(W), B, K, N, ACT[act], DROP_KIND[drop_kind],
                                       drop_p, seed, site, ptr(ctx.seed_word), ptr(dpre), ptr(dx), ptr(dW), ptr(db),
                                       stream_ptr()), "mmf_dense_backward")
        return dx, dW, db, None, None, None, None, None


def dense(x, W, b, act="none", drop_kind="none", drop_p=0.0, seed=0, site=0):
    return DenseFn.apply(x, W, b, act, drop_kind, drop_p, seed, site)


class GateMulFn(torch.autograd.Function):
    """sigmoid(z) * h (models/model_modules.py:163)."""

    @staticmethod
    def forward(ctx, z, h):
        z, h = _f32c(z), _f32c(h)
        o = torch.empty_like(h)
        check(lib().mmf_gate_mul_forward(ptr(z), ptr(h), ptr(o), h.numel(), stream_ptr()), "mmf_gate_mul_forward")
        ctx.save_for_backward(z, h)
        return o

    @staticmethod
    def backward(ctx, g):
        z, h = ctx.saved_tensors
        g = _f32c(g)
        dz, dh = torch.empty_like(z), torch.empty_like(h)
        check(lib().mmf_gate_mul_backward(ptr(g), ptr(z), ptr(h), ptr(dz), ptr(dh), h.numel(), stream_ptr()),
              "mmf_gate_mul_backward")
        return dz, dh


def gate_mul(z, h):
    return GateMulFn.apply(z, h)


class KronFn(torch.autograd.Function):
    """[o0,1] (x) [o1,1] ((x) [o2,1]) + post-fusion Dropout (models/model_modules.py:164-171)."""

    @staticmethod
    def forward(ctx, drop_p, seed, site, *os_):
        os_ = [_f32c(o) for o in os_]
        m = len(os_)
        B, dim = os_[0].shape
        total = (dim + 1) ** m
        out = torch.empty((B, total), dtype=torch.float32, device=os_[0].device)
        arr = (C.c_void_p * m)(*[ptr(o) for o in os_])
        word = _seed_word
        check(lib().mmf_kron_forward(arr, m, dim, B, float(drop_p), int(seed) & 0xFFFFFFFF, int(site), ptr(word), ptr(out),
                                     stream_ptr()), "mmf_kron_forward")
        ctx.cfg = (float(drop_p), int(seed) & 0xFFFFFFFF, int(site))
        ctx.seed_word = word
        ctx.save_for_backward(*os_)
        return out

    @staticmethod
    def backward(ctx, g):
        os_ = list(ctx.saved_tensors)
        drop_p, seed, site = ctx.cfg
        m = len(os_)
        B, dim = os_[0].shape
        g = _f32c(g)
        ds = [torch.empty_like(o) for o in os_]
        arr = (C.c_void_p * m)(*[ptr(o) for o in os_])
        darr = (C.c_void_p * m)(*[ptr(d) for d in ds])
        check(lib().mmf_kron_backward(ptr(g), arr, m, dim, B, drop_p, seed, site, ptr(ctx.seed_word), darr, stream_ptr()),
              "mmf_kron_backward")
        return (None, None, None) + tuple(ds)


def kron_ones(os_, drop_p=0.0, seed=0, site=0):
    return KronFn.apply(drop_p, seed, site, *os_)


# ---- raw (no-autograd) launch helpers shared by the fused fusion Function --------------------------------------
def _dense_fwd_raw(x, W, b, act, kind, p, seed, site, word=None, out=None):
    B, K = x.shape
    N = W.shape[0]
    y = torch.empty((B, N), dtype=torch.float32, device=x.device) if out is None else out
    check(lib().mmf_dense_forward(ptr(x), ptr(W), ptr(b), B, K, N, ACT[act], DROP_KIND[kind], float(p), seed, site,
                                  ptr(word), ptr(y), stream_ptr()), "mmf_dense_forward")
    return y


def _dense_bwd_raw(gy, y, x, W, has_bias, act, kind, p, seed, site, need_dx=True, word=None):
    B, K = x.shape
    N = W.shape[0]
    dpre = torch.empty_like(y)
    dx = torch.empty_like(x) if need_dx else None
    dW = torch.empty_like(W)
    db = torch.empty((N,), dtype=torch.float32, device=x.device) if has_bias else None
    check(lib().mmf_dense_backward(ptr(gy), ptr(y), ptr(x), ptr(W), B, K, N, ACT[act], DROP_KIND[kind], float(p), seed,
                                   site, ptr(word), ptr(dpre), ptr(dx), ptr(dW), ptr(db), stream_ptr()), "mmf_dense_backward")
    return dx, dW, db


class MlpFn(torch.autograd.Function):
    """A chain of dense layers as ONE autograd node (the omic head: SNN blocks + classifier,
    models/model_genomic.py:56-72): same dense kernels, one Python forward / backward instead of one per layer.
    spec: tuple of (act, drop_kind, drop_p, site) per layer; tensors: x, then (W, b) per layer."""

    @staticmethod
    def forward(ctx, spec, seed, x, *wb):
        x = _f32c(x)
        seed = int(seed) & 0xFFFFFFFF
        acts = [x]
        word = _seed_word
        for i, (act, kind, p, site) in enumerate(spec):
            acts.append(_dense_fwd_raw(acts[-1], wb[2 * i], wb[2 * i + 1], act, kind, p, seed, site, word))
        ctx.cfg = (spec, seed)
        ctx.seed_word = word
        ctx.save_for_backward(*acts, *wb)
        return acts[-1]

    @staticmethod
    def backward(ctx, g):
        spec, seed = ctx.cfg
        n = len(spec)
        t = ctx.saved_tensors
        acts, wb = t[:n + 1], t[n + 1:]
        g = _f32c(g)
        grads = [None] * (2 * n)
        for i in range(n - 1, -1, -1):
            act, kind, p, site = spec[i]
            need_dx = i > 0 or ctx.needs_input_grad[2]
            g, dW, db = _dense_bwd_raw(g, acts[i + 1], acts[i], wb[2 * i], wb[2 * i + 1] is not None, act, kind, p, seed,
                                       site, need_dx=need_dx, word=ctx.seed_word)
            grads[2 * i], grads[2 * i + 1] = dW, db
        return (None, None, g) + tuple(grads)


def mlp(x, layers, seed=0):
    """layers: list of (W, b, act, drop_kind, drop_p, site)."""
    spec = tuple((a, k, float(p), int(s)) for (_, _, a, k, p, s) in layers)
    wb = []
    for (W, b, *_r) in layers:
        wb += [W, b]
    return MlpFn.apply(spec, seed, x, *wb)


class XFusionFn(torch.autograd.Function):
    """The whole XlinearFusion block (models/model_modules.py:156-178, gate=1, skip=1) as ONE autograd node: the same
    HIP kernels as the composable ops above, but one Python forward and one Python backward instead of ~25 nodes
    (the block is latency-bound; at B = 1 the autograd/ctypes dispatch dominated it).

    params: per modality (Wh, bh, Wz, bz, Wo, bo), then We1, be1, We2, be2.  Dropout sites: o_i -> i, post-fusion -> 8,
    encoder1 -> 9, encoder2 -> 10 (same as the composable path)."""

    @staticmethod
    def forward(ctx, m, p, seed, *tensors):
        vs = [_f32c(t) for t in tensors[:m]]
        w = [_f32c(t) for t in tensors[m:]]
        kind = "dropout" if p > 0 else "none"
        seed = int(seed) & 0xFFFFFFFF
        B = vs[0].shape[0]
        sdim = w[0].shape[0]
        new = lambda: [torch.empty((B, sdim), dtype=torch.float32, device=vs[0].device) for _ in range(m)]
        hs, zs, gms, os_ = new(), new(), new(), new()
        io = XFusionFn._io(m, vs, w, hs, zs, gms, os_)
        word = _seed_word
        check(lib().mmf_xreduce_forward(C.byref(io), float(p), seed, ptr(word), stream_ptr()), "mmf_xreduce_forward")
        We1, be1, We2, be2 = w[6 * m:6 * m + 4]
        B, dim = os_[0].shape
        kr = torch.empty((B, (dim + 1) ** m), dtype=torch.float32, device=vs[0].device)
        arr = (C.c_void_p * m)(*[ptr(o) for o in os_])
        check(lib().mmf_kron_forward(arr, m, dim, B, float(p), seed, 8, ptr(word), ptr(kr), stream_ptr()), "mmf_kron_forward")
        e1 = _dense_fwd_raw(kr, We1, be1, "relu", kind, p, seed, 9, word)
        cat2 = torch.cat([e1] + vs, dim=1)
        e2 = _dense_fwd_raw(cat2, We2, be2, "relu", kind, p, seed, 10, word)
        ctx.cfg = (m, float(p), seed, kind)
        ctx.seed_word = word
        ctx.save_for_backward(*vs, *w, *hs, *zs, *gms, *os_, kr, e1, cat2, e2)
        return e2

    @staticmethod
    def _io(m, vs, w, hs, zs, gms, os_):
        io = _lib.XReduceIO(m=m, B=vs[0].shape[0], dim=vs[0].shape[1], sdim=w[0].shape[0])
        for i in range(m):
            Wh, bh, Wz, bz, Wo, bo = w[6 * i:6 * i + 6]
            io.v[i] = ptr(vs[i]); io.Wh[i] = ptr(Wh); io.bh[i] = ptr(bh); io.Wz[i] = ptr(Wz); io.bz[i] = ptr(bz)
            io.Wo[i] = ptr(Wo); io.bo[i] = ptr(bo)
            io.h[i] = ptr(hs[i]); io.z[i] = ptr(zs[i]); io.gm[i] = ptr(gms[i]); io.o[i] = ptr(os_[i])
        return io

    @staticmethod
    def backward(ctx, g):
        m, p, seed, kind = ctx.cfg
        t = list(ctx.saved_tensors)
        vs, t = t[:m], t[m:]
        w, t = t[:6 * m + 4], t[6 * m + 4:]
        hs, zs, gms, os_ = t[:m], t[m:2 * m], t[2 * m:3 * m], t[3 * m:4 * m]
        kr, e1, cat2, e2 = t[4 * m:4 * m + 4]
        We1, be1, We2, be2 = w[6 * m:6 * m + 4]
        g = _f32c(g)
        word = ctx.seed_word
        d_cat2, dWe2, dbe2 = _dense_bwd_raw(g, e2, cat2, We2, True, "relu", kind, p, seed, 10, word=word)
        H1 = e1.shape[1]
        d_e1 = d_cat2[:, :H1].contiguous()
        dim_v = vs[0].shape[1]
        d_kr, dWe1, dbe1 = _dense_bwd_raw(d_e1, e1, kr, We1, True, "relu", kind, p, seed, 9, word=word)
        B, dim = os_[0].shape
        d_os = [torch.empty_like(o) for o in os_]
        arr = (C.c_void_p * m)(*[ptr(o) for o in os_])
        darr = (C.c_void_p * m)(*[ptr(d) for d in d_os])
        check(lib().mmf_kron_backward(ptr(d_kr), arr, m, dim, B, p, seed, 8, ptr(word), darr, stream_ptr()), "mmf_kron_backward")
        io = XFusionFn._io(m, vs, w, hs, zs, gms, os_)
        dvs = [torch.empty_like(v) for v in vs]
        grads_w = []
        for i in range(m):
            Wh, bh, Wz, bz, Wo, bo = w[6 * i:6 * i + 6]
            gw = [torch.empty_like(x) for x in (Wh, bh, Wz, bz, Wo, bo)]
            io.d_o[i] = ptr(d_os[i]); io.dv[i] = ptr(dvs[i])
            io.dWh[i], io.dbh[i], io.dWz[i], io.dbz[i], io.dWo[i], io.dbo[i] = [ptr(x) for x in gw]
            grads_w += gw
        check(lib().mmf_xreduce_backward(C.byref(io), p, seed, ptr(word), stream_ptr()), "mmf_xreduce_backward")
        for i in range(m):      # skip connection: encoder2 saw the v_i directly
            dvs[i] += d_cat2[:, H1 + dim_v * i: H1 + dim_v * (i + 1)]
        return (None, None, None) + tuple(dvs) + tuple(grads_w) + (dWe1, dbe1, dWe2, dbe2)


def xfusion(v_list, weights, p=0.0, seed=0):
    """weights: [Wh_0, bh_0, Wz_0, bz_0, Wo_0, bo_0, ..., We1, be1, We2, be2]."""
    return XFusionFn.apply(len(v_list), p, seed, *v_list, *weights)


# ---- stage-2 building blocks (include/mmf_amil.h "Stage-2 building blocks") -------------------------
class BatchNormFn(torch.autograd.Function):
    """y = dropout(act(BatchNorm1d(x) [+ res])) in one launch; running statistics are updated in place (training)."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, running_mean, running_var, training, eps, momentum, act, drop_p, seed, site):
        x = _f32c(x)
        res = _f32c(res) if res is not None else None
        B, F = x.shape
        y = torch.empty_like(x)
        mean = torch.empty((F,), dtype=torch.float32, device=x.device)
        invstd = torch.empty_like(mean)
        word = _seed_word
        check(lib().mmf_batchnorm_forward(ptr(x), ptr(res), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                                          B, F, 1 if training else 0, float(eps), float(momentum), ACT[act],
                                          float(drop_p), int(seed) & 0xFFFFFFFF, int(site), ptr(word), ptr(y), ptr(mean),
                                          ptr(invstd), stream_ptr()), "mmf_batchnorm_forward")
        ctx.cfg = (bool(training), act, float(drop_p), int(seed) & 0xFFFFFFFF, int(site), res is not None)
        ctx.seed_word = word
        ctx.save_for_backward(x, y, gamma, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, gamma, mean, invstd = ctx.saved_tensors
        training, act, drop_p, seed, site, has_res = ctx.cfg
        gy = _f32c(gy)
        B, F = x.shape
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if has_res else None
        dgamma = torch.empty_like(mean)
        dbeta = torch.empty_like(mean)
        check(lib().mmf_batchnorm_backward(ptr(gy), ptr(y), ptr(x), ptr(gamma), ptr(mean), ptr(invstd), B, F,
                                           1 if training else 0, ACT[act], drop_p, seed, site, ptr(ctx.seed_word),
                                           ptr(dx), ptr(dres), ptr(dgamma), ptr(dbeta), stream_ptr()),
              "mmf_batchnorm_backward")
        return dx, dres, (dgamma if gamma is not None else None), (dbeta if gamma is not None else None), \
            None, None, None, None, None, None, None, None, None


def batchnorm(x, bn, res=None, act="none", drop_p=0.0, seed=0, site=0):
    """Apply an nn.BatchNorm1d module's parameters / buffers on the GPU (train or eval as bn.training says)."""
    training = bn.training or bn.running_mean is None
    if bn.training and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    mom = 0.0 if bn.momentum is None else bn.momentum
    return BatchNormFn.apply(x, res, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.eps, mom, act,
                             drop_p, seed, site)


class HighwayMixFn(torch.autograd.Function):
    """models/model_modules.py:21-25: sigmoid(zg) * relu(zn) + (1 - sigmoid(zg)) * zl."""

    @staticmethod
    def forward(ctx, zg, zn, zl):
        zg, zn, zl = _f32c(zg), _f32c(zn), _f32c(zl)
        y = torch.empty_like(zg)
        check(lib().mmf_highway_mix_forward(ptr(zg), ptr(zn), ptr(zl), zg.numel(), ptr(y), stream_ptr()),
              "mmf_highway_mix_forward")
        ctx.save_for_backward(zg, zn, zl)
        return y

    @staticmethod
    def backward(ctx, gy):
        zg, zn, zl = ctx.saved_tensors
        gy = _f32c(gy)
        dzg, dzn, dzl = torch.empty_like(zg), torch.empty_like(zg), torch.empty_like(zg)
        check(lib().mmf_highway_mix_backward(ptr(gy), ptr(zg), ptr(zn), ptr(zl), zg.numel(), ptr(dzg), ptr(dzn), ptr(dzl),
                                             stream_ptr()), "mmf_highway_mix_backward")
        return dzg, dzn, dzl


def highway_mix(zg, zn, zl):
    return HighwayMixFn.apply(zg, zn, zl)


class RankLossFn(torch.autograd.Function):
    """utils/loss_utils.py:58-101; loss and d(risks) in one launch."""

    @staticmethod
    def forward(ctx, risks, times, c, phi, reduction):
        shape = risks.shape
        r = _f32c(risks.reshape(-1))
        B = r.numel()
        if B == 1:
            raise NotImplementedError("Batch size must be at least 2")          # as the reference (loss_utils.py:60-61)
        t = torch.as_tensor(times).reshape(B).to(device=r.device, dtype=torch.float64).contiguous()
        cc = c.reshape(B).to(device=r.device, dtype=torch.float32).contiguous()
        loss = torch.empty((), dtype=torch.float32, device=r.device)
        dr = torch.empty_like(r)
        check(lib().mmf_ranking_loss(ptr(r), ptr(t), ptr(cc), B, {"sigmoid": 0, "relu": 1}[phi],
                                     {"mean": 0, "sum": 1}[reduction], ptr(loss), ptr(dr), stream_ptr()), "mmf_ranking_loss")
        ctx.save_for_backward(dr)
        ctx.shape = shape
        return loss

    @staticmethod
    def backward(ctx, g):
        (dr,) = ctx.saved_tensors
        return (dr * g).reshape(ctx.shape), None, None, None, None


def ranking_loss(risks, times, c, phi="sigmoid", reduction="mean"):
    return RankLossFn.apply(risks, times, c, phi, reduction)


class HazardFn(torch.autograd.Function):
    """logits -> (risk, hazards, S, Y_hat): models/nll_models_pretrained.py:58-62."""

    @staticmethod
    def forward(ctx, logits):
        logits = _f32c(logits)
        B, K = logits.shape
        hz, S = torch.empty_like(logits), torch.empty_like(logits)
        risk = torch.empty((B,), dtype=torch.float32, device=logits.device)
        Y_hat = torch.empty((B, 1), dtype=torch.int64, device=logits.device)
        check(lib().mmf_hazards_forward(ptr(logits), B, K, ptr(hz), ptr(S), ptr(Y_hat), ptr(risk), stream_ptr()),
              "mmf_hazards_forward")
        ctx.save_for_backward(hz)
        ctx.mark_non_differentiable(Y_hat)
        return risk, hz, S, Y_hat

    @staticmethod
    def backward(ctx, g_risk, g_hz, g_S, _gY):
        (hz,) = ctx.saved_tensors
        B, K = hz.shape
        f = lambda t: _f32c(t) if t is not None else None
        dl = torch.empty_like(hz)
        check(lib().mmf_hazards_backward(ptr(f(g_hz)), ptr(f(g_S)), ptr(f(g_risk)), ptr(hz), B, K, ptr(dl), stream_ptr()),
              "mmf_hazards_backward")
        return dl


def hazards_from_logits(logits):
    return HazardFn.apply(logits)
