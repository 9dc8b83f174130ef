"""ctypes binding of libmmf_amil.so (C ABI: include/mmf_amil.h).

There is no CPU fallback: if the shared library is missing or a call is made with
non-GPU tensors the functions raise.  The library itself is only *loaded* here (symbol
resolution); no HIP call happens until an entry point is invoked.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MMF_LIB_PATH") or os.path.join(_HERE, "libmmf_amil.so")   # override: diagnostic builds only

ABI_VERSION = 11

c_f32p = C.c_void_p   # device pointers travel as integers (tensor.data_ptr())


class AmilDesc(C.Structure):
    """struct mmf_amil_desc (include/mmf_amil.h)."""
    _fields_ = [
        ("N", C.c_int64), ("L", C.c_int32), ("H", C.c_int32), ("D", C.c_int32), ("gated", C.c_int32),
        ("W1", C.c_void_p), ("b1", C.c_void_p), ("Wa", C.c_void_p), ("ba", C.c_void_p),
        ("Wb", C.c_void_p), ("bb", C.c_void_p), ("Wc", C.c_void_p), ("bc", C.c_void_p),
        ("p_h", C.c_float), ("p_att", C.c_float), ("seed", C.c_uint32),
        ("seed_dev", C.c_void_p), ("trace", C.c_void_p), ("concurrent", C.c_int32), ("gemm", C.c_int32),
        ("sync", C.c_void_p), ("sync_words", C.c_int32),
    ]


class AmilGrads(C.Structure):
    """struct mmf_amil_grads (include/mmf_amil.h)."""
    _fields_ = [
        ("dW1", C.c_void_p), ("db1", C.c_void_p), ("dWa", C.c_void_p), ("dba", C.c_void_p),
        ("dWb", C.c_void_p), ("dbb", C.c_void_p), ("dWc", C.c_void_p), ("dbc", C.c_void_p),
        ("dx", C.c_void_p),
    ]


class SurvHead(C.Structure):
    """struct mmf_surv_head (include/mmf_amil.h)."""
    _fields_ = [("Wk", C.c_void_p), ("bk", C.c_void_p), ("K", C.c_int32), ("logits", C.c_void_p),
                ("hazards", C.c_void_p), ("S", C.c_void_p), ("Y_hat", C.c_void_p), ("risk", C.c_void_p)]


class NllTarget(C.Structure):
    """struct mmf_nll_target (include/mmf_amil.h)."""
    _fields_ = [("Y", C.c_void_p), ("c", C.c_void_p), ("alpha", C.c_float), ("eps", C.c_float),
                ("loss_scale", C.c_float), ("loss", C.c_void_p), ("dWk", C.c_void_p), ("dbk", C.c_void_p),
                ("accumulate", C.c_int32)]


class MaxnetDesc(C.Structure):
    """struct mmf_maxnet_desc (include/mmf_amil.h)."""
    _fields_ = [("B", C.c_int32), ("G", C.c_int32), ("H0", C.c_int32), ("H1", C.c_int32),
                ("x", C.c_void_p), ("W0", C.c_void_p), ("b0", C.c_void_p), ("W1", C.c_void_p), ("b1", C.c_void_p),
                ("Wc", C.c_void_p), ("bc", C.c_void_p), ("p_drop", C.c_float), ("seed", C.c_uint32),
                ("seed_dev", C.c_void_p), ("sync", C.c_void_p), ("sync_words", C.c_int32), ("trace", C.c_void_p)]


class MaxnetGrads(C.Structure):
    """struct mmf_maxnet_grads (include/mmf_amil.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("dW0", "db0", "dW1", "db1", "dWc", "dbc")]


class XReduceIO(C.Structure):
    """struct mmf_xreduce_io (include/mmf_amil.h)."""
    _P3 = C.c_void_p * 3
    _fields_ = [("m", C.c_int32), ("B", C.c_int32), ("dim", C.c_int32), ("sdim", C.c_int32)] + [
        (n, C.c_void_p * 3) for n in ("v", "Wh", "bh", "Wz", "bz", "Wo", "bo", "h", "z", "gm", "o", "d_o", "dv",
                                      "dWh", "dbh", "dWz", "dbz", "dWo", "dbo")]


# name -> (restype, argtypes): every symbol include/mmf_amil.h declares
SYMBOLS = {
    "mmf_strerror": (C.c_char_p, [C.c_int]),
    "mmf_abi_version": (C.c_int, []),
    "mmf_amil_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "mmf_amil_forward": (C.c_int, [C.POINTER(AmilDesc), C.c_void_p, C.c_void_p, C.c_size_t,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    "mmf_amil_backward": (C.c_int, [C.POINTER(AmilDesc), C.c_void_p, C.c_void_p, C.c_size_t,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.POINTER(AmilGrads), C.c_void_p]),
    "mmf_amil_bf16_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "mmf_amil_bf16_forward": (C.c_int, [C.POINTER(AmilDesc), C.c_void_p, C.c_void_p, C.c_size_t,
                                        C.c_void_p, C.c_void_p, C.c_void_p]),
    "mmf_amil_bf16_backward": (C.c_int, [C.POINTER(AmilDesc), C.c_void_p, C.c_void_p, C.c_size_t,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.POINTER(AmilGrads), C.c_void_p]),
    "mmf_amil_head_forward": (C.c_int, [C.POINTER(AmilDesc), C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t,
                                        C.POINTER(SurvHead), C.c_void_p, C.c_void_p, C.c_void_p]),
    "mmf_amil_nll_step": (C.c_int, [C.POINTER(AmilDesc), C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t,
                                    C.POINTER(SurvHead), C.POINTER(NllTarget), C.c_void_p, C.POINTER(AmilGrads), C.c_void_p]),
    "mmf_surv_head_nll_step": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(SurvHead), C.POINTER(NllTarget), C.c_void_p,
                                         C.c_void_p]),
    "mmf_amil_infer_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "mmf_amil_infer": (C.c_int, [C.POINTER(AmilDesc), C.c_void_p, C.c_void_p, C.c_size_t,
                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    "mmf_amil_bf16_infer_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "mmf_amil_bf16_infer": (C.c_int, [C.POINTER(AmilDesc), C.c_void_p, C.c_void_p, C.c_size_t,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "mmf_attn_net_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "mmf_attn_net_forward": (C.c_int, [C.POINTER(AmilDesc), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "mmf_attn_net_backward": (C.c_int, [C.POINTER(AmilDesc), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                        C.POINTER(AmilGrads), C.c_void_p]),
    "mmf_linear_forward_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "mmf_linear_forward": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int64,
                                     C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                     C.c_float, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_size_t, C.c_void_p, C.c_int32, C.c_void_p]),
    "mmf_maxnet_cox_step_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "mmf_maxnet_cox_step": (C.c_int, [C.POINTER(MaxnetDesc), C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_size_t,
                                      C.c_void_p, C.c_void_p, C.POINTER(MaxnetGrads), C.c_int32, C.c_void_p]),
    "mmf_linear_backward_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "mmf_linear_backward": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int64,
                                      C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_size_t, C.c_void_p]),
    "mmf_surv_head_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mmf_surv_head_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int32, C.c_int32, C.c_int32,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mmf_nll_surv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                               C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mmf_cox_surv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                               C.c_void_p, C.c_void_p, C.c_void_p]),
    "mmf_dropout_keep_host": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_float]),
    "mmf_dense_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_int32, C.c_float, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    "mmf_dense_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_float, C.c_uint32, C.c_uint32, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mmf_gate_mul_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "mmf_gate_mul_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int32, C.c_void_p]),
    "mmf_kron_forward": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int32,
                                   C.c_float, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mmf_kron_backward": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int32,
                                    C.c_float, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p]),
    "mmf_adam_l1_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                                   C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int32, C.c_void_p]),
    "mmf_abs_sum": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mmf_xreduce_forward": (C.c_int, [C.POINTER(XReduceIO), C.c_float, C.c_uint32, C.c_void_p, C.c_void_p]),
    "mmf_xreduce_backward": (C.c_int, [C.POINTER(XReduceIO), C.c_float, C.c_uint32, C.c_void_p, C.c_void_p]),
    "mmf_batchnorm_forward": (C.c_int, [C.c_void_p] * 6 + [C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32,
                                                         C.c_float, C.c_uint32, C.c_uint32] + [C.c_void_p] * 5),
    "mmf_batchnorm_backward": (C.c_int, [C.c_void_p] * 6 + [C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                          C.c_float, C.c_uint32, C.c_uint32] + [C.c_void_p] * 6),
    "mmf_highway_mix_forward": (C.c_int, [C.c_void_p] * 3 + [C.c_int64, C.c_void_p, C.c_void_p]),
    "mmf_highway_mix_backward": (C.c_int, [C.c_void_p] * 4 + [C.c_int64] + [C.c_void_p] * 4),
    "mmf_ranking_loss": (C.c_int, [C.c_void_p] * 3 + [C.c_int32] * 3 + [C.c_void_p] * 3),
    "mmf_hazards_forward": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 5),
    "mmf_hazards_backward": (C.c_int, [C.c_void_p] * 4 + [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "mmf_trace_create": (C.c_void_p, [C.c_int32]),
    "mmf_trace_destroy": (None, [C.c_void_p]),
    "mmf_trace_dump": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
}
# exported by the diagnostic builds only (tools/diag_build.py, -DMMF_STAMPS)
DIAG_SYMBOLS = {
    "mmf_debug_stamps": (None, [C.c_int, C.POINTER(C.c_uint64)]),
}


class KernelTrace:
    """Per-kernel device time of the attention-stack calls made inside the `with` block (include/mmf_amil.h
    "Kernel trace": HIP events on the launch stream).  dump() -> {kernel_name: (launches, total_ms)}."""

    def __init__(self, capacity: int = 4096):
        self.handle = lib().mmf_trace_create(capacity)
        if not self.handle:
            raise MmfError("mmf_trace_create failed")

    def __enter__(self):
        from . import ops
        self._prev = ops.set_trace(self.handle)
        return self

    def __exit__(self, *a):
        from . import ops
        ops.set_trace(self._prev)

    def dump(self):
        l = lib()
        buf = C.create_string_buffer(1 << 16)     # one call: the dump clears the records
        l.mmf_trace_dump(self.handle, buf, 1 << 16)
        out = {}
        for line in buf.value.decode().splitlines():
            name, cnt, ms = line.split()
            out[name] = (int(cnt), float(ms))
        return out

    def close(self):
        if self.handle:
            lib().mmf_trace_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

_lib = None


class MmfError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load libmmf_amil.so once; raise loudly if it is missing (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MmfError(
                f"{LIB_PATH} not found: the HIP extension has not been built. "
                "Run `python -m multimodalfusion_amd.build` (needs hipcc, gfx950 target). "
                "There is no CPU fallback for the attention-MIL path.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        for name, (res, args) in DIAG_SYMBOLS.items():
            if hasattr(l, name):
                fn = getattr(l, name)
                fn.restype = res
                fn.argtypes = args
        v = l.mmf_abi_version()
        if v != ABI_VERSION:
            raise MmfError(f"libmmf_amil.so ABI version {v}, binding expects {ABI_VERSION}: rebuild")
        _lib = l
    return _lib


def check(code: int, what: str):
    if code != 0:
        msg = lib().mmf_strerror(code).decode()
        raise MmfError(f"{what} failed: {msg} (code {code})")


def ptr(t):
    """Device pointer of a contiguous fp32/int64/fp64 CUDA(HIP) tensor, or None."""
    if t is None:
        return None
    if not t.is_cuda:
        raise MmfError("multimodalfusion_amd ops need tensors on the MI355X (device 'cuda'); "
                       "there is no CPU path -- call model.relocate() / .to('cuda') first")
    if not t.is_contiguous():
        raise MmfError("tensor must be contiguous")
    return t.data_ptr()


def stream_ptr():
    """The raw handle of torch's current HIP stream on the current device.  (torch.cuda.current_stream() builds a Stream
    object and goes through is_available(), which reads the environment: ~5 us a call, twenty calls in a multimodal step.)"""
    import torch
    torch.cuda._lazy_init()
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
