"""Bag feed: host -> HBM staging of the next bag overlapped with the current bag's kernels.

The reference copies every tensor of a batch synchronously at the top of each iteration
(utils/core_utils.py:194-198: `.to(device)` on pageable memory).  At ~1 ms of GPU work per 50k bag the 205 MB copy
(~3.3 ms over PCIe Gen5 x16) dominates, so the feed keeps `depth` bags in flight: pinned host staging buffers, copies
issued with non_blocking=True on a dedicated HIP stream, an event per bag that the compute stream waits on.

`DevicePrefetcher(loader)` wraps any iterable that yields the reference's batch tuple
(radio_features: dict, path_features, genomic_features, label, event_time, c) -- e.g. a DataLoader built with
collate_MIL_survival (utils/utils.py:35-46) -- and yields the same tuple with the tensors already on the GPU.
"""
from __future__ import annotations

from collections import deque

import torch


def load_slide_bags(paths, pin: bool = True, dtype=None) -> torch.Tensor:
    """One subject's pathology bag from its per-slide `.pt` files, as datasets/dataset_survival.py:359-366 builds it
    (`torch.load` each slide, `torch.cat(dim=0)`), but assembled directly in ONE pinned host buffer: the slides are
    read into their row ranges, so there is no pageable intermediate and the H2D copy can be asynchronous.
    An empty list gives the reference's "missing" sentinel `torch.zeros((1, 1))` (dataset_survival.py:356-357).
    `dtype` (e.g. torch.bfloat16) narrows while copying -- meant for bags already stored in that type on disk, where
    it is a no-op; narrowing 200 MB of fp32 on the host costs more than the PCIe time it saves."""
    if len(paths) == 0:
        return torch.zeros((1, 1))
    bags = [torch.load(p, map_location="cpu") for p in paths]
    for b in bags:
        if b.dim() != 2 or b.shape[1] != bags[0].shape[1]:
            raise ValueError("slide bags must be [n_i x L] with one feature width")
    dt = dtype or bags[0].dtype
    out = torch.empty((sum(b.shape[0] for b in bags), bags[0].shape[1]), dtype=dt,
                      pin_memory=bool(pin) and torch.cuda.is_available())
    r = 0
    for b in bags:
        out[r:r + b.shape[0]].copy_(b)
        r += b.shape[0]
    return out


def intersect_modalities(features_by_mod, slice_index_by_mod, modalities=None, pin: bool = True, dtype=None,
                         require_equal: bool = True):
    """One subject's radiology bags restricted to the slices every modality has, as datasets/dataset_survival.py:346-348
    does it: `intersect = set.intersection(*[set(v) ...])`, then per modality the rows whose slice index is in the
    intersection, IN THEIR STORED ORDER (`features[np.in1d(slice_index, intersect), :]`).  Pure index work: the rows
    are gathered bit for bit.

    features_by_mod[m]: [n_m x 1024] array or tensor, slice_index_by_mod[m]: [n_m] slice ids (any integer or float type).
    `modalities` fixes the order of the returned dict (default: the order of `features_by_mod`; the reference iterates
    `self.modalities`).  Returns {m: tensor [n x 1024]} in ONE pinned host buffer per modality (what DevicePrefetcher copies
    asynchronously).  The model concatenates the modalities along the feature axis (models/model_attention_mil_radio.py:
    80-82), which needs equal n; a slice id that repeats inside one modality breaks that in the reference too (torch.cat
    raises there), so `require_equal` (default) raises ValueError here, at the point where the cause is still known."""
    import numpy as np
    mods = list(modalities) if modalities is not None else list(features_by_mod.keys())
    if not mods:
        return {}
    idx = {m: np.asarray(slice_index_by_mod[m]).reshape(-1) for m in mods}
    common = None
    for m in mods:
        u = np.unique(idx[m])
        common = u if common is None else np.intersect1d(common, u, assume_unique=True)
    out = {}
    for m in mods:
        f = features_by_mod[m]
        f = f if torch.is_tensor(f) else torch.as_tensor(np.asarray(f))
        if f.dim() != 2 or f.shape[0] != idx[m].shape[0]:
            raise ValueError(f"modality {m}: features {tuple(f.shape)} do not match {idx[m].shape[0]} slice ids")
        rows = torch.as_tensor(np.nonzero(np.isin(idx[m], common))[0])
        dst = torch.empty((rows.numel(), f.shape[1]), dtype=dtype or f.dtype,
                          pin_memory=bool(pin) and torch.cuda.is_available())
        if rows.numel():
            torch.index_select(f, 0, rows, out=dst) if dst.dtype == f.dtype else dst.copy_(f.index_select(0, rows))
        out[m] = dst
    if require_equal and len({t.shape[0] for t in out.values()}) > 1:
        raise ValueError("modalities keep different numbers of slices (a slice id repeats inside one modality): "
                         + ", ".join(f"{m}: {t.shape[0]}" for m, t in out.items()))
    return out


def load_radio_bags(h5_paths_by_mod, modalities=None, pin: bool = True, dtype=None):
    """datasets/dataset_survival.py:339-348 for one subject: read `features` and `slice_index` of every modality's .h5
    file, keep the common slices.  Needs h5py (absent from the build image: the read is three lines and untested here; the
    index work, which is what must be exact, is `intersect_modalities`)."""
    try:
        import h5py
    except ImportError as e:      # fail loudly, as everything on this path does
        raise ImportError("load_radio_bags needs h5py; pass arrays to feed.intersect_modalities instead") from e
    feats, idx = {}, {}
    for m, path in h5_paths_by_mod.items():
        with h5py.File(path, "r") as f:
            feats[m] = f["features"][:]
            idx[m] = f["slice_index"][:]
    return intersect_modalities(feats, idx, modalities, pin, dtype)


def _pin(t: torch.Tensor) -> torch.Tensor:
    if not torch.is_tensor(t) or t.is_cuda:
        return t
    return t if t.is_pinned() else t.pin_memory()


class DevicePrefetcher:
    def __init__(self, loader, device=None, depth: int = 2, path_dtype=None):
        """path_dtype=torch.bfloat16 delivers the pathology bag in bf16, which selects the bf16-storage kernels
        (include/mmf_amil.h: mmf_amil_bf16_*).  A bag that is already bf16 on the host crosses PCIe at half the
        bytes; an fp32 bag is copied as it is and narrowed on the device, on the copy stream."""
        self.loader = loader
        self.path_dtype = path_dtype
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.depth = max(1, int(depth))
        self.stream = torch.cuda.Stream(self.device)

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        radio, path, genomic, label, event_time, c = batch
        with torch.cuda.stream(self.stream):
            move = lambda t: _pin(t).to(self.device, non_blocking=True) if torch.is_tensor(t) else t
            path_d = move(path)
            if self.path_dtype is not None and torch.is_tensor(path_d) and path_d.dim() == 2 and path_d.shape[1] > 1 \
                    and path_d.dtype != self.path_dtype:
                path_d = path_d.to(self.path_dtype)          # on the copy stream, after the H2D of the fp32 bag
            out = ({k: move(v) for k, v in radio.items()}, path_d,
                   move(genomic.float() if torch.is_tensor(genomic) else genomic), move(label), event_time, move(c))
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return out, ev

    def __iter__(self):
        it = iter(self.loader)
        q = deque()
        try:
            for _ in range(self.depth):
                q.append(self._stage(next(it)))
        except StopIteration:
            pass
        while q:
            out, ev = q.popleft()
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)                       # compute waits for THIS bag only; later copies keep flowing
            for t in list(out[0].values()) + [out[1], out[2], out[3], out[5]]:
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(cur)             # allocator: do not reuse before the compute stream is done
            try:
                q.append(self._stage(next(it)))
            except StopIteration:
                pass
            yield out


# ---- rank sharding of the training loader (one-bag-per-GPU data parallelism) ------------------------------------
class _StridedBatches(torch.utils.data.Sampler):
    """Batch sampler of rank r: every world-th batch of the base batch sampler's order for this epoch.  The order is
    drawn ONCE, on rank 0, and broadcast, so the ranks agree on it by construction (a RandomSampler draws its
    permutation from the global torch RNG, which nothing guarantees to be in the same state on every rank)."""

    def __init__(self, base, rank, world):
        self.base, self.rank, self.world = base, rank, world
        self.n_total = len(base)

    def __iter__(self):
        import torch.distributed as dist
        order = [list(b) for b in self.base] if self.rank == 0 else None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            box = [order]
            dist.broadcast_object_list(box, src=0)
            order = box[0]
        elif order is None:
            order = [list(b) for b in self.base]
        self.n_total = len(order)
        return iter(order[self.rank::self.world])

    def __len__(self):
        return (self.n_total - self.rank + self.world - 1) // self.world


class RankShard:
    """The part of `loader` that rank `rank` of `world` processes: loader positions rank, rank + world, ... --
    only those bags are read from disk, collated and copied.  Iterating yields the reference's batch tuples in
    position order; `n_total` is the length of the whole (unsharded) loader, `position(i)` the loader position of the
    i-th yielded batch.  Works for a torch DataLoader (a new DataLoader over the same dataset whose batch sampler is
    the strided view of the original one), for a DevicePrefetcher around one, for indexable sequences, and for any
    other sized iterable (there the skipped items are still produced by the iterable, then dropped)."""

    def __init__(self, loader, rank: int, world: int):
        self.rank, self.world = int(rank), int(world)
        self.n_total = len(loader)
        self._src = self._shard(loader)

    def _shard(self, loader):
        r, w = self.rank, self.world
        if isinstance(loader, DevicePrefetcher):
            return DevicePrefetcher(self._shard(loader.loader), loader.device, loader.depth, loader.path_dtype)
        if isinstance(loader, torch.utils.data.DataLoader):
            kw = dict(collate_fn=loader.collate_fn, num_workers=loader.num_workers, pin_memory=loader.pin_memory,
                      timeout=loader.timeout, worker_init_fn=loader.worker_init_fn)
            if loader.num_workers > 0:
                kw.update(prefetch_factor=loader.prefetch_factor, persistent_workers=loader.persistent_workers)
            return torch.utils.data.DataLoader(loader.dataset, batch_sampler=_StridedBatches(loader.batch_sampler, r, w), **kw)
        if hasattr(loader, "__getitem__"):
            return (loader[i] for i in range(r, self.n_total, w))
        import itertools
        return itertools.islice(iter(loader), r, None, w)

    def position(self, i: int) -> int:
        return self.rank + i * self.world

    def __iter__(self):
        return iter(self._src)
