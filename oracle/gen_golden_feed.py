"""Fixtures for feed.intersect_modalities: runs the reference's OWN three lines on seeded inputs.

The lines (datasets/dataset_survival.py:346-348: the set intersection of the modalities' slice ids and the np.in1d row
selection) sit inside a Dataset.__getitem__ that needs h5py and the study's CSVs, neither of which this image has, so the
script reads exactly those source lines from /root/reference AS TEXT at generation time, dedents them and executes them in
a namespace holding `radio_features`, `slices_index`, `self.modalities`, `np`, `torch`.  Nothing of the reference is kept in
this repository: the fixture is inputs and outputs only.

    python oracle/gen_golden_feed.py        # -> tests/golden/feed.npz   (build container only: needs /root/reference)
"""
import json
import os
import sys
import textwrap
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MMF_REFERENCE", "/root/reference")
LINES = (346, 348)     # 1-based, inclusive


def reference_lines():
    src = open(os.path.join(REF, "datasets", "dataset_survival.py")).read().splitlines()
    body = "\n".join(src[LINES[0] - 1:LINES[1]])
    assert "set.intersection" in body and "np.in1d" in body, "reference lines moved"
    return textwrap.dedent(body)


def cases():
    rng = np.random.default_rng(20261004)
    mods = ["T1", "T2", "T1Gd", "FLAIR"]
    out = []

    def make(name, ids_by_mod, width=24, dtype=np.float32):     # the gather does not care about the width: keep the fixture small
        feats = {m: rng.standard_normal((len(v), width)).astype(dtype) for m, v in ids_by_mod.items()}
        out.append((name, list(ids_by_mod.keys()), feats, {m: np.asarray(v) for m, v in ids_by_mod.items()}))

    make("four_mods_partial_overlap", {m: np.sort(rng.choice(60, size=n, replace=False)).astype(np.int64)
                                       for m, n in zip(mods, (40, 35, 44, 38))})
    make("unsorted_ids", {m: rng.permutation(np.arange(10, 10 + n)).astype(np.int32) for m, n in zip(mods, (20, 25, 18, 22))})
    make("single_modality", {"T1": np.arange(7, dtype=np.int64)})
    make("empty_intersection", {"T1": np.arange(0, 5), "T2": np.arange(5, 10)})
    make("float_ids", {"T1": np.array([0.0, 1.0, 2.5, 4.0]), "T2": np.array([2.5, 4.0, 7.0]), "FLAIR": np.array([4.0, 2.5, 0.0])}, width=16)
    make("identical_ids_full_width", {m: np.arange(12, dtype=np.int64) for m in mods}, width=1024)
    make("repeated_id_inside_a_modality", {"T1": np.array([1, 2, 2, 3]), "T2": np.array([2, 3, 4])}, width=8)
    return out


def main():
    code = reference_lines()
    arrays, meta = {}, {"reference_lines": f"datasets/dataset_survival.py:{LINES[0]}-{LINES[1]}", "cases": []}
    for name, mods, feats, idx in cases():
        ns = {"np": np, "torch": torch, "self": types.SimpleNamespace(modalities=mods),
              "radio_features": {m: feats[m].copy() for m in mods}, "slices_index": {m: idx[m].copy() for m in mods}}
        exec(code, ns)
        meta["cases"].append({"name": name, "modalities": mods})
        for m in mods:
            arrays[f"{name}/in/{m}/features"] = feats[m]
            arrays[f"{name}/in/{m}/slice_index"] = idx[m]
            arrays[f"{name}/out/{m}"] = ns["radio_features"][m].numpy()
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(ROOT, "tests", "golden", "feed.npz")
    np.savez_compressed(path, **arrays)
    print(path, os.path.getsize(path), "bytes;", len(meta["cases"]), "cases")


if __name__ == "__main__":
    sys.exit(main())
