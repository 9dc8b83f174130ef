"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Repo-owned deterministic generator for synthetic bags and weights, so that the
GPU box can rebuild exactly the inputs the golden fixtures were made from
without the reference being present and without depending on torch's RNG
streams.  Counter-based: value i of a stream is a pure function of
(seed, stream, i).

Shapes / key names follow SURVEY.md Appendix B (state_dict contract of
models/model_attention_mil_path.py:13-34, model_attention_mil_radio.py:14-51,
model_genomic.py:13-39, model_mm_attention_mil.py:19-98 in the reference).
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

_U = np.uint64


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + _U(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> _U(30))) * _U(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U(27))) * _U(0x94D049BB133111EB)
        return z ^ (z >> _U(31))


def uniform01(seed: int, n: int, stream: int = 0) -> np.ndarray:
    """n float64 values in the open interval (0, 1)."""
    with np.errstate(over="ignore"):
        key = _splitmix64(np.array([seed], dtype=np.uint64) * _U(0xD1342543DE82EF95)
                          + _U(stream) * _U(0x2545F4914F6CDD1D))
        z = _splitmix64(np.arange(n, dtype=np.uint64) ^ key)
    return ((z >> _U(11)).astype(np.float64) + 0.5) / float(1 << 53)


def normal(seed: int, shape, stream: int = 0, std: float = 1.0, dtype=np.float32) -> np.ndarray:
    """i.i.d. N(0, std^2) via Box-Muller on two uniform streams."""
    shape = tuple(int(s) for s in np.atleast_1d(shape))
    n = int(np.prod(shape)) if shape else 1
    m = (n + 1) // 2
    u1 = uniform01(seed, m, 2 * stream)
    u2 = uniform01(seed, m, 2 * stream + 1)
    r = np.sqrt(-2.0 * np.log(u1))
    z = np.empty(2 * m, dtype=np.float64)
    z[0::2] = r * np.cos(2.0 * np.pi * u2)
    z[1::2] = r * np.sin(2.0 * np.pi * u2)
    return (z[:n] * std).reshape(shape).astype(dtype)


def bag(seed: int, n_inst: int, dim: int = 1024, stream: int = 0, nonneg: bool = False) -> np.ndarray:
    """Synthetic [n_inst x dim] fp32 feature bag, N(0,1).

    nonneg=True gives |N(0,1)| -- closer to real post-ReLU pooled ResNet features.
    """
    x = normal(seed, (n_inst, dim), stream=stream)
    return np.abs(x) if nonneg else x


# --------------------------------------------------------------------------
# state-dict builders
# --------------------------------------------------------------------------

def _linear(sd, name, out_f, in_f, seed, stream, init, bias_std):
    if init == "xavier":          # utils/utils.py:217-222 (xavier_normal_, zero bias)
        std = float(np.sqrt(2.0 / (in_f + out_f)))
    elif init == "max":           # utils/utils.py:228-233 (N(0, 1/sqrt(fan_in)), zero bias)
        std = 1.0 / float(np.sqrt(in_f))
    else:
        raise ValueError(init)
    sd[name + ".weight"] = normal(seed, (out_f, in_f), stream=stream, std=std)
    if bias_std > 0:
        sd[name + ".bias"] = normal(seed, (out_f,), stream=stream + 1, std=bias_std)
    else:
        sd[name + ".bias"] = np.zeros((out_f,), dtype=np.float32)
    return stream + 2


SIZE_DICT = {"small": [1024, 256, 256], "big": [1024, 512, 384]}
# MM model uses its own table (model_mm_attention_mil.py:28-30): big = [1024, 256, 384]
SIZE_DICT_MM = {"small": [1024, 256, 256], "big": [1024, 256, 384]}
SIZE_DICT_OMIC = {"small": [256, 256], "big": [1024, 256]}


def _attn_stack(sd, prefix, size, gated, dropout, seed, stream, init, bias_std):
    """Sequential(Linear, ReLU, Dropout, Attn_Net[_Gated]) -- key names per Appendix B."""
    L, H, D = size
    stream = _linear(sd, f"{prefix}.0", H, L, seed, stream, init, bias_std)
    if gated:
        stream = _linear(sd, f"{prefix}.3.attention_a.0", D, H, seed, stream, init, bias_std)
        stream = _linear(sd, f"{prefix}.3.attention_b.0", D, H, seed, stream, init, bias_std)
        stream = _linear(sd, f"{prefix}.3.attention_c", 1, D, seed, stream, init, bias_std)
    else:
        # Attn_Net: module = [Linear, Tanh, (Dropout), Linear]; last index is 3 with dropout else 2
        stream = _linear(sd, f"{prefix}.3.module.0", D, H, seed, stream, init, bias_std)
        last = 3 if dropout else 2
        stream = _linear(sd, f"{prefix}.3.module.{last}", 1, D, seed, stream, init, bias_std)
    return stream


def path_state_dict(seed=1, gated=True, size="small", n_classes=4, dropout=False, bias_std=0.0):
    sd = OrderedDict()
    sz = SIZE_DICT[size]
    st = _attn_stack(sd, "attention_net_WSI", sz, gated, dropout, seed, 0, "xavier", bias_std)
    _linear(sd, "classifier", n_classes, sz[1], seed, st, "xavier", bias_std)
    return sd


def radio_state_dict(seed=1, gated=True, n_classes=4, dropout=True, n_mod=4, bias_std=0.0):
    sd = OrderedDict()
    sz = SIZE_DICT["small"]  # model_attention_mil_radio.py:70 forces 'small'
    st = 0
    if n_mod > 1:
        st = _linear(sd, "reduce_dim", sz[0], sz[0] * n_mod, seed, st, "xavier", bias_std)
    st = _attn_stack(sd, "attention_net_radio", sz, gated, dropout, seed, st, "xavier", bias_std)
    _linear(sd, "classifier", n_classes, sz[1], seed, st, "xavier", bias_std)
    return sd


def maxnet_state_dict(seed=1, input_dim=36, size="small", nll=True, n_classes=4, bias_std=0.0):
    sd = OrderedDict()
    hid = SIZE_DICT_OMIC[size]
    st = _linear(sd, "fc_omic.0.0", hid[0], input_dim, seed, 0, "max", bias_std)
    for i in range(1, len(hid)):
        st = _linear(sd, f"fc_omic.{i}.0", hid[i], hid[i - 1], seed, st, "max", bias_std)
    _linear(sd, "classifier", n_classes if nll else 1, hid[-1], seed, st, "max", bias_std)
    return sd


def mm_state_dict(seed=1, input_dim=80, fusion="concat", gate_path=True, gate_radio=True,
                  dropout=False, n_classes=4, mode="radio_path_omic", n_mod=4,
                  size_wsi="small", size_omic="small", bias_std=0.0):
    """Key order follows the submodule creation order in model_mm_attention_mil.py:34-95."""
    sd = OrderedDict()
    so = SIZE_DICT_OMIC[size_omic]
    sr = SIZE_DICT_MM["small"]
    sw = SIZE_DICT_MM[size_wsi]
    st = _linear(sd, "fc_omic.0.0", so[0], input_dim, seed, 0, "xavier", bias_std)
    for i in range(1, len(so)):
        st = _linear(sd, f"fc_omic.{i}.0", so[i], so[i - 1], seed, st, "xavier", bias_std)
    st = _attn_stack(sd, "attention_net_radio", sr, gate_radio, dropout, seed, st, "xavier", bias_std)
    st = _linear(sd, "reduce_dim", sr[0], sr[0] * n_mod, seed, st, "xavier", bias_std)
    st = _attn_stack(sd, "attention_net_WSI", sw, gate_path, dropout, seed, st, "xavier", bias_std)
    n_fused = sum(k in mode for k in ("radio", "path", "omic"))
    if fusion == "tensor":
        dim, sdim = 256, 16
        for i in range(n_fused):
            st = _linear(sd, f"mm.reduce.{i}.0.0", sdim, dim, seed, st, "xavier", bias_std)
            st = _linear(sd, f"mm.reduce.{i}.1.0", sdim, dim * n_fused, seed, st, "xavier", bias_std)
            st = _linear(sd, f"mm.reduce.{i}.2.0", sdim, sdim, seed, st, "xavier", bias_std)
        st = _linear(sd, "mm.encoder1.0", 512, (sdim + 1) ** n_fused, seed, st, "xavier", bias_std)
        st = _linear(sd, "mm.encoder2.0", 512, 512 + dim * n_fused, seed, st, "xavier", bias_std)
        st = _linear(sd, "classifier.0", 256, 512, seed, st, "xavier", bias_std)
        st = _linear(sd, "classifier.3", n_classes, 256, seed, st, "xavier", bias_std)
    else:
        csize = 0
        if "radio" in mode:
            csize += sr[1]
        if "path" in mode:
            csize += sw[1]
        if "omic" in mode:
            csize += so[1]
        st = _linear(sd, "classifier", n_classes, csize, seed, st, "xavier", bias_std)
    return sd


# --------------------------------------------------------------------------
# dropout masks: the SAME 32-bit counter hash the HIP kernels use
# (multimodalfusion_amd/csrc/mmf_common.h: mmf_keep()).  Integer arithmetic,
# bit-exact by construction, so train-mode parity can be checked exactly.
# --------------------------------------------------------------------------

def keep_mask(seed: int, site: int, rows: int, cols: int, p: float) -> np.ndarray:
    """Boolean keep-mask [rows x cols]; element index = row*cols+col (uint32 wrap)."""
    idx = (np.arange(rows, dtype=np.uint64)[:, None] * _U(cols)
           + np.arange(cols, dtype=np.uint64)[None, :]).astype(np.uint32)
    key = np.uint32((int(seed) + 0x632BE5AB * (int(site) + 1)) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        h = idx * np.uint32(0x9E3779B1) + key
        h ^= h >> np.uint32(16)
        h *= np.uint32(0x85EBCA6B)
        h ^= h >> np.uint32(13)
        h *= np.uint32(0xC2B2AE35)
        h ^= h >> np.uint32(16)
    thr = np.uint32(int(p * (1 << 24)))
    return (h >> np.uint32(8)) >= thr


def drop_scale_mask(seed, site, rows, cols, p, dtype=np.float32):
    """Mask already scaled by 1/(1-p) as nn.Dropout applies it in train mode."""
    return keep_mask(seed, site, rows, cols, p).astype(dtype) / dtype(1.0 - p)
