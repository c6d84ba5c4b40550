"""Packed training steps on disk and their streaming into HBM.

The reference rebuilds every batch of every epoch in Python: DataProcessor.generate_batch_reactions
(data/load_reactions.py:336-421) -> Parsing_features.parsing_reactions (:574-577) -> BatchMolGraph list appends
(features/featurization.py:246-288), about 0.35 ms per molecule, then five H2D tensor copies inside every forward
(models/mpn.py:77).  Here a dataset is packed ONCE by the native packer into a shard file and an epoch is a stream of
page-cache reads + one pinned H2D copy per step that runs on a copy stream underneath the previous step's compute:

  ShardWriter(path).add_step(r_batch, p_batch, scope, targets, add) ... .close()
  reader = ShardReader(path)                       # np.memmap, zero-copy views per step
  for step in StepPrefetcher(reader, device, order=range(len(reader))):
      out = model(step["r"], step["p"], gpu, step["add"]);  loss(out, step["scope"], step["targets"], gpu) ...

File layout (little endian; every step blob starts on a 4096-byte boundary, every array inside it on 256 bytes, so a
blob maps 1:1 onto a pinned staging buffer and a device buffer and every array is a typed view of it):

  [0, 64)      header: magic "RRSHARD1", u32 version, u32 n_steps, u64 index_offset, u32 atom_fdim, u32 bond_fdim
  step blobs   i64 n_sections, i64 blob_bytes, then n_sections x (i64 key_id, i64 rows, i64 cols (0 = 1-D), i64 offset), arrays
  index        n_steps x 8 i64: offset, nbytes, Q, M, nA_p, nB_p, K, has_unique

What a step carries (int32 tables, float32 features):
  p.*   product graph: f_atoms [nA,64], fbond [nB,24] (ONLY the 22 bond columns of f_bonds - the 61 atom columns are
        f_atoms[b2a] (featurization.py:198-199) and are rebuilt on the device, rr_build_fbonds_f32), a2b, b2a, b2revb,
        a2a, a_scope and the backward tables
  r.*   reactant graph: tables only when the reactants repeat (every candidate of a query carries the same reactant,
        train_listwise.py:188); its features are gathers of the distinct reactants' rows through amap / bmap
  u.*   the distinct reactants (full graph), amap / amap_t / bmap / bmap_t (BatchMolGraph.unique())
  scope [Q] int32, targets [M] float32, add [M,F] float32
"""
from __future__ import annotations

import os
import queue
import struct
import threading
import time
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .featurization import ATOM_FDIM, BOND_FDIM, BatchMolGraph, DeviceBatch, DeviceGraph

MAGIC = b"RRSHARD1"
VERSION = 1
HEADER_BYTES = 64
BLOB_ALIGN = 4096
ARRAY_ALIGN = 256

_TABLES = ("a2b", "b2a", "b2revb", "a2a", "a_scope", "a2b_rev_t", "b2t", "a2a_t", "npad", "atom2mol", "b2b_t", "npad_b")
_FLOAT = {"f_atoms", "fbond", "npad", "npad_b", "targets", "add"}
_KEYS: List[str] = []
for _side in ("p", "r", "u"):
    _KEYS += [f"{_side}.f_atoms", f"{_side}.fbond"] + [f"{_side}.{k}" for k in _TABLES]
_KEYS += ["amap", "amap_t", "bmap", "bmap_t", "scope", "targets", "add"]
_KEY_ID = {k: i for i, k in enumerate(_KEYS)}


def _dtype_of(key: str):
    return np.float32 if key.split(".")[-1] in _FLOAT else np.int32


def _align(n: int, a: int) -> int:
    return (n + a - 1) // a * a


def _side_arrays(side: str, batch: BatchMolGraph, with_features: bool) -> Dict[str, np.ndarray]:
    h = batch._host
    out = {}
    if with_features:
        afd, bfd = h["atom_fdim"], h["bond_fdim"]
        out[f"{side}.f_atoms"] = h["f_atoms"]
        nb = bfd - afd                                            # bond-only columns, padded to a 16-byte row
        fb = np.zeros((h["nB"], _align(nb, 4)), np.float32)
        fb[:, :nb] = h["f_bonds"][:, afd:bfd]
        out[f"{side}.fbond"] = fb
    for k in _TABLES:
        out[f"{side}.{k}"] = h[k]
    return out


class ShardWriter:
    """Append packed steps to a shard file (see the module docstring for the layout)."""

    def __init__(self, path: str):
        self.path = path
        self.f = open(path, "wb")
        self.f.write(b"\0" * BLOB_ALIGN)                          # header page, patched by close()
        self.index: List[List[int]] = []
        self.atom_fdim, self.bond_fdim = ATOM_FDIM, BOND_FDIM

    def add_step(self, r_batch: BatchMolGraph, p_batch: BatchMolGraph, scope: Sequence[int], targets, add=None) -> None:
        if r_batch.n_atoms != p_batch.n_atoms or r_batch.n_mols != p_batch.n_mols:
            raise ValueError("reactant and product batches must hold the same molecules / atoms (base_model.py:168)")
        if r_batch.max_num_bonds != p_batch.max_num_bonds:
            raise ValueError("pack both sides with the same pad width K (hazard H1): BatchMolGraph(..., K=K)")
        arrs = _side_arrays("p", p_batch, True)
        ub, amap, amap_t = r_batch.unique()
        has_u = ub.n_mols < r_batch.n_mols
        arrs.update(_side_arrays("r", r_batch, not has_u))
        if has_u:
            bmap, bmap_t = r_batch.unique_bonds()
            arrs.update(_side_arrays("u", ub, True))
            arrs.update(amap=amap, amap_t=amap_t, bmap=bmap, bmap_t=bmap_t)
        arrs["scope"] = np.asarray(scope, np.int32)
        arrs["targets"] = np.asarray(targets, np.float32).reshape(-1)
        if add is not None:
            arrs["add"] = np.asarray(add, np.float32).reshape(p_batch.n_mols, -1)
        toc, off = [], _align(16 + 32 * len(arrs), ARRAY_ALIGN)
        for k, a in arrs.items():
            a = np.ascontiguousarray(a, _dtype_of(k))
            arrs[k] = a
            rows, cols = (a.shape[0], a.shape[1]) if a.ndim == 2 else (a.shape[0], 0)      # cols == 0: 1-D
            toc.append((_KEY_ID[k], rows, cols, off))
            off = _align(off + a.nbytes, ARRAY_ALIGN)
        nbytes = _align(off, BLOB_ALIGN)
        start = self.f.tell()
        assert start % BLOB_ALIGN == 0
        blob = bytearray(nbytes)
        struct.pack_into("<qq", blob, 0, len(toc), nbytes)
        for i, t in enumerate(toc):
            struct.pack_into("<qqqq", blob, 16 + 32 * i, *t)
        for (kid, rows, cols, o), a in zip(toc, arrs.values()):
            blob[o:o + a.nbytes] = a.tobytes()
        self.f.write(blob)
        hp = p_batch._host
        self.index.append([start, nbytes, len(arrs["scope"]), p_batch.n_mols, hp["nA"], hp["nB"], hp["K"], int(has_u)])
        self.atom_fdim, self.bond_fdim = hp["atom_fdim"], hp["bond_fdim"] - hp["atom_fdim"]

    def close(self) -> None:
        idx_off = self.f.tell()
        self.f.write(np.asarray(self.index, np.int64).reshape(-1, 8).tobytes())
        self.f.seek(0)
        self.f.write(MAGIC + struct.pack("<IIQII", VERSION, len(self.index), idx_off, self.atom_fdim, self.bond_fdim))
        self.f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


class ShardReader:
    """Memory-mapped shard file: `host_step(i)` gives zero-copy numpy views, `blob(i)` the step's raw bytes."""

    def __init__(self, path: str):
        self.path = path
        self.mm = np.memmap(path, dtype=np.uint8, mode="r")
        head = bytes(self.mm[:HEADER_BYTES])
        if head[:8] != MAGIC:
            raise RuntimeError(f"{path}: not a reactranker shard file")
        ver, n, idx_off, self.atom_fdim, self.bond_fdim = struct.unpack_from("<IIQII", head, 8)
        if ver != VERSION:
            raise RuntimeError(f"{path}: shard version {ver}, this build reads {VERSION}")
        self.index = np.frombuffer(self.mm, np.int64, n * 8, idx_off).reshape(n, 8)
        self.max_step_bytes = int(self.index[:, 1].max()) if n else 0

    def __len__(self) -> int:
        return self.index.shape[0]

    def meta(self, i: int) -> dict:
        off, nbytes, Q, M, nA, nB, K, has_u = (int(v) for v in self.index[i])
        return dict(offset=off, nbytes=nbytes, Q=Q, M=M, nA=nA, nB=nB, K=K, has_unique=bool(has_u))

    def blob(self, i: int) -> np.ndarray:
        off, nbytes = int(self.index[i, 0]), int(self.index[i, 1])
        return self.mm[off:off + nbytes]

    @staticmethod
    def toc(blob: np.ndarray) -> Dict[str, tuple]:
        """name -> (rows, cols, byte offset inside the blob)"""
        n = int(np.frombuffer(blob, np.int64, 1, 0)[0])
        t = np.frombuffer(blob, np.int64, 4 * n, 16).reshape(n, 4)
        return {_KEYS[int(k)]: (int(r), int(c), int(o)) for k, r, c, o in t}

    def host_step(self, i: int) -> Dict[str, np.ndarray]:
        blob = self.blob(i)
        out = {}
        for k, (rows, cols, off) in self.toc(blob).items():
            a = np.frombuffer(blob, _dtype_of(k), rows * max(1, cols), off)
            out[k] = a.reshape(rows, cols) if cols else a
        return out


class ShardSet:
    """Several shard files read as one sequence of steps (a dataset is usually written by several packer processes)."""

    toc = staticmethod(ShardReader.toc)

    def __init__(self, paths: Sequence[str]):
        self.readers = [ShardReader(p) for p in paths]
        if not self.readers:
            raise ValueError("ShardSet needs at least one shard file")
        self.atom_fdim, self.bond_fdim = self.readers[0].atom_fdim, self.readers[0].bond_fdim
        for r in self.readers:
            if (r.atom_fdim, r.bond_fdim) != (self.atom_fdim, self.bond_fdim):
                raise RuntimeError("shard files with different feature widths")
        self.index = np.concatenate([r.index for r in self.readers], 0)
        self._where = [(ri, i) for ri, r in enumerate(self.readers) for i in range(len(r))]
        self.max_step_bytes = max(r.max_step_bytes for r in self.readers)

    def __len__(self) -> int:
        return len(self._where)

    def largest(self, k: int = 2) -> List[int]:
        """the k steps with the most atoms + bonds: run these first (untimed / as the first steps of an epoch) and every
        buffer the allocator has to find for a later step already exists - a fresh hipMalloc in the middle of an epoch is
        a 40-180 ms step"""
        size = self.index[:, 4] + self.index[:, 5]
        return [int(i) for i in np.argsort(size, kind="stable")[::-1][:max(0, k)]]

    def meta(self, i: int) -> dict:
        ri, j = self._where[i]
        return self.readers[ri].meta(j)

    def blob(self, i: int) -> np.ndarray:
        ri, j = self._where[i]
        return self.readers[ri].blob(j)

    def host_step(self, i: int) -> Dict[str, np.ndarray]:
        ri, j = self._where[i]
        return self.readers[ri].host_step(j)


def _typed_views(buf: torch.Tensor, toc: Dict[str, tuple]) -> Dict[str, torch.Tensor]:
    """Typed tensor views of a uint8 buffer holding one step blob (host or device)."""
    out = {}
    for k, (rows, cols, off) in toc.items():
        dt = torch.float32 if _dtype_of(k) == np.float32 else torch.int32
        v = buf[off:off + rows * max(1, cols) * 4].view(dt)
        out[k] = v.view(rows, cols) if cols else v
    return out


def _gather_rows(src: torch.Tensor, idx: torch.Tensor, width: int) -> torch.Tensor:
    """out[r] = src[idx[r], 0:width] (device): a K = 1 gather-sum."""
    from . import functions as Fn
    out = torch.empty(idx.shape[0], src.shape[1], dtype=torch.float32, device=src.device)
    return Fn.gather_sum(src, idx.reshape(-1, 1), width, out=out)


def device_step(views: Dict[str, torch.Tensor], meta: dict, device, atom_fdim: int = ATOM_FDIM, bond_fdim: int = BOND_FDIM):
    """Batch objects over the typed device views of one step blob (no copies except the reactant feature gathers)."""
    def graph(side, n_mols, extra=None):
        t = {k: views[f"{side}.{k}"] for k in _TABLES}
        for k in ("f_atoms", "fbond"):
            if f"{side}.{k}" in views:
                t[k] = views[f"{side}.{k}"]
        if extra:
            t.update(extra)
        nA, K = t["a2b"].shape
        return DeviceGraph.from_device(t, nA, t["b2a"].shape[0], K, n_mols, device, atom_fdim, bond_fdim)
    M = meta["M"]
    pg = graph("p", M)
    if meta["has_unique"]:
        ug = graph("u", views["u.a_scope"].shape[0])
        amap, bmap = views["amap"], views["bmap"]
        # the full reactant batch repeats the distinct reactants' feature rows
        rg = graph("r", M, dict(f_atoms=_gather_rows(ug.f_atoms, amap, ug.f_atoms.shape[1]),
                                fbond=_gather_rows(ug.fbond, bmap, ug.fbond.shape[1])))
        rb = DeviceBatch(rg, DeviceBatch(ug), amap, views["amap_t"], bmap, views["bmap_t"])
    else:
        rb = DeviceBatch(graph("r", M))
    return rb, DeviceBatch(pg)


class StepPrefetcher:
    """Iterate over the steps of a shard with the next `depth - 1` steps already on their way into HBM.

    A reader thread copies step blobs from the memory map into pinned staging buffers (page-cache reads, the GIL is
    released inside the copy) and issues ONE H2D copy per step on a dedicated copy stream; the consumer's stream waits
    on that copy's event only.  A slot (pinned + device buffer) is recycled after the consumer has moved on by `depth`
    steps; the copy stream then waits for an event the consumer recorded when it released the slot, so a step's
    buffers are never overwritten while its kernels may still be running."""

    def __init__(self, reader: ShardReader, device, order: Iterable[int], depth: int = 3, copy_threads: int = 4):
        if depth < 2:
            raise ValueError("depth >= 2 (one step in use, at least one in flight)")
        self.reader, self.device, self.depth = reader, torch.device(device), depth
        self.order = list(order)
        nb = _align(max(1, reader.max_step_bytes), BLOB_ALIGN)
        self.pinned = [torch.empty(nb, dtype=torch.uint8).pin_memory() for _ in range(depth)]
        self.pinned_np = [t.numpy() for t in self.pinned]
        self.dev = [torch.empty(nb, dtype=torch.uint8, device=self.device) for _ in range(depth)]
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.ready: "queue.Queue" = queue.Queue()
        self.free: "queue.Queue" = queue.Queue()
        for s in range(depth):
            self.free.put((s, None, None))                       # (slot, release event of its last user, its H2D event)
        self._held: List[list] = []                              # [slot, H2D event, end-of-step event]
        self._err: Optional[BaseException] = None
        self._stop = False
        self.bytes_copied = 0
        self.wait_s = 0.0
        self.copy_threads = copy_threads
        self._copiers = ThreadPoolExecutor(max_workers=max(1, copy_threads), thread_name_prefix="rr-shard-copy")
        self.thread = threading.Thread(target=self._run, name="rr-shard-reader", daemon=True)
        self.thread.start()

    def _run(self):
        try:
            torch.cuda.set_device(self.device)
            for i in self.order:
                slot, released, last_copy = self.free.get()
                if self._stop:
                    return
                if last_copy is not None:
                    last_copy.synchronize()                      # the pinned buffer's previous H2D has been read out
                blob = self.reader.blob(i)
                n = blob.shape[0]
                self._stage(self.pinned_np[slot], blob, n)       # page cache -> pinned staging (numpy drops the GIL)
                with torch.cuda.stream(self.copy_stream):
                    if released is not None:
                        self.copy_stream.wait_event(released)    # the slot's last user's kernels have finished
                    self.dev[slot][:n].copy_(self.pinned[slot][:n], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self.copy_stream)
                self.bytes_copied += n
                self.ready.put((slot, i, ev))
            self.ready.put(None)
        except BaseException as e:                               # surfaced to the consumer
            self._err = e
            self.ready.put(None)

    def _stage(self, dst: np.ndarray, src: np.ndarray, n: int) -> None:
        """One step blob (~55 MB) into its pinned slot, cut into a few pieces copied in parallel: a single memcpy thread
        moves ~10 GB/s, i.e. one blob per ~5.5 ms - the training step's own duration - so one thread alone leaves the
        input pipeline no margin."""
        parts = self.copy_threads
        if parts <= 1 or n < (8 << 20):
            np.copyto(dst[:n], src[:n])
            return
        step = _align((n + parts - 1) // parts, BLOB_ALIGN)
        futs = [self._copiers.submit(np.copyto, dst[a:min(a + step, n)], src[a:min(a + step, n)]) for a in range(0, n, step)]
        for f in futs:
            f.result()

    def prime(self) -> float:
        """Block until the first step's H2D copy has been issued and completed (the pipeline's start-up latency: first
        page-cache touches, first use of the pinned slots, first DMA out of them); returns the seconds waited.  Optional -
        iteration works without it."""
        t0 = time.perf_counter()
        while self.ready.empty() and self._err is None and self.thread.is_alive():
            time.sleep(0.0005)
        # ... and has LANDED: the first DMA out of freshly pinned slots can take tens of milliseconds (measured: 37-95 ms
        # showing up in the first consumer step when only the issue was awaited)
        if self._err is not None:                                # the reader died before it staged anything: say so here,
            raise RuntimeError("shard reader thread failed") from self._err   # not one silent slow step later
        with self.ready.mutex:                                   # (queue.Queue's own lock: the reader may be appending)
            first = self.ready.queue[0] if self.ready.queue else None
        if first is not None and first[2] is not None:
            first[2].synchronize()
        return time.perf_counter() - t0

    def _views(self, slot: int, i: int) -> Dict[str, torch.Tensor]:
        return _typed_views(self.dev[slot], ShardReader.toc(self.reader.blob(i)))

    def __iter__(self):
        return self

    def __next__(self):
        # the consumer has finished ENQUEUEING the step it was handed last: an event recorded now on its stream marks
        # the end of that step's kernels.  A slot goes back to the reader one call later, tagged with ITS step's event.
        if self._held:
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(self.device))
            self._held[-1][2] = done
        while len(self._held) >= self.depth - 1:
            slot, h2d, done = self._held.pop(0)
            self.free.put((slot, done, h2d))
        t0 = time.perf_counter()
        item = self.ready.get()
        self.wait_s += time.perf_counter() - t0                   # time the consumer stood waiting for the reader thread
        if item is None:
            if self._err is not None:
                raise RuntimeError("shard reader thread failed") from self._err
            raise StopIteration
        slot, i, ev = item
        torch.cuda.current_stream(self.device).wait_event(ev)
        meta = self.reader.meta(i)
        views = self._views(slot, i)
        rb, pb = device_step(views, meta, self.device, self.reader.atom_fdim, self.reader.bond_fdim)
        self._held.append([slot, ev, None])
        host = self.reader.host_step(i)
        return dict(r=rb, p=pb, scope=[int(v) for v in host["scope"]], targets=views["targets"],
                    add=views.get("add"), index=i, meta=meta)

    def close(self):
        self._stop = True
        while not self.free.empty():
            try:
                self.free.get_nowait()
            except queue.Empty:
                break
        for s in range(self.depth):
            self.free.put((s, None, None))
        self.thread.join(timeout=5)
        self._copiers.shutdown(wait=False)


def load_step(reader: ShardReader, i: int, device) -> dict:
    """One step, synchronously (tests / small jobs): same objects as StepPrefetcher yields."""
    dev = torch.device(device)
    buf = torch.from_numpy(np.array(reader.blob(i))).to(dev)     # private copy; the views below alias it on the device
    views = _typed_views(buf, ShardReader.toc(reader.blob(i)))
    meta = reader.meta(i)
    rb, pb = device_step(views, meta, dev, reader.atom_fdim, reader.bond_fdim)
    host = reader.host_step(i)
    return dict(r=rb, p=pb, scope=[int(v) for v in host["scope"]], targets=views["targets"], add=views.get("add"),
                index=i, meta=meta)
