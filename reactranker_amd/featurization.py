"""Graph batching for the HIP path — host-side mirror of the reference's
`reactranker/features/featurization.py` (BatchMolGraph :231-335, MolGraph :135-210).

`BatchMolGraph` keeps the reference's attributes and accessors (`get_components()`,
`get_a2a()`, `a_scope`, `max_num_bonds`, ...) but is built by the native packer
(`rr_pack_graphs`, csrc/pack.cpp) instead of Python list appends, uses int32 indices and
16-byte aligned feature rows, and also carries the transposed index tables the backward
kernels need.  `DeviceGraph` is its resident-in-HBM form; it is uploaded once per batch
object and cached (the reference re-copies five tensors per forward, models/mpn.py:77).

Molecules enter as `synth.MolSpec` arrays, as any object with the reference MolGraph's attributes, or - when RDKit
is installed - as SMILES through the optional featuriser in rdkit_features.py.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .synth import ATOM_FDIM, BOND_FDIM, MolSpec

FA_LD = 64     # f_atoms row stride (61 -> 64 floats, 16-byte aligned rows)
FB_LD = 84     # f_bonds row stride (83 -> 84)


def get_atom_fdim() -> int:
    return ATOM_FDIM


def get_bond_fdim() -> int:
    return BOND_FDIM


class MolGraph:
    """Single-molecule graph (reference featurization.py:135-210) backed by arrays.

    `MolGraph(smiles)` featurises through RDKit when the package is installed (rdkit_features.py: the reference's
    feature layout, atom order and bond numbering) and raises a RuntimeError that says so when it is not;
    `MolGraph.from_spec(spec)` wraps arrays that already exist.
    """

    def __init__(self, smiles: str = None, reaction: bool = True, atom_messages: bool = False, *, spec: MolSpec = None):
        if spec is None:
            if smiles is None:
                raise ValueError("MolGraph needs a SMILES / InChI string or spec=MolSpec")
            if atom_messages:
                raise ValueError("atom_messages=True drops the atom features from f_bonds; the encoder (bond_fdim 83) "
                                 "never uses that form")
            from . import rdkit_features
            spec = rdkit_features.spec_from_smiles(smiles, reaction)
        self.spec = spec
        self.smiles = spec.smiles if smiles is None else smiles
        self.n_atoms = spec.n_atoms
        self.n_bonds = spec.n_bonds

    @classmethod
    def from_spec(cls, spec: MolSpec) -> "MolGraph":
        return cls(spec=spec)


def _concat_specs(specs: Sequence[MolSpec]):
    """Vectorised MolGraph construction for a whole batch of MolSpecs (no per-bond python)."""
    M = len(specs)
    mol_atoms = np.fromiter((s.n_atoms for s in specs), dtype=np.int32, count=M)
    n_edges = np.fromiter((s.edges.shape[0] for s in specs), dtype=np.int64, count=M)
    mol_bonds = (2 * n_edges).astype(np.int32)
    f_atoms_cat = np.concatenate([s.f_atoms for s in specs], axis=0).astype(np.float32, copy=False) if M else \
        np.zeros((0, ATOM_FDIM), np.float32)
    edges = np.concatenate([s.edges for s in specs], axis=0).astype(np.int64) if M else np.zeros((0, 2), np.int64)
    fbond = np.concatenate([s.f_bond for s in specs], axis=0).astype(np.float32, copy=False) if M else \
        np.zeros((0, BOND_FDIM), np.float32)
    E = edges.shape[0]
    atom_off = np.concatenate([[0], np.cumsum(mol_atoms, dtype=np.int64)])
    edge_off = np.concatenate([[0], np.cumsum(n_edges)])
    mol_of_edge = np.repeat(np.arange(M), n_edges)
    e_local = np.arange(E) - edge_off[mol_of_edge]
    # directed bonds, reference numbering: b1 = a1->a2 (even), b2 = a2->a1 (odd)  featurization.py:202-210
    b2a_local = np.empty(2 * E, np.int32)
    b2a_local[0::2] = edges[:, 0]
    b2a_local[1::2] = edges[:, 1]
    b2revb_local = np.empty(2 * E, np.int32)
    b2revb_local[0::2] = 2 * e_local + 1
    b2revb_local[1::2] = 2 * e_local
    src_global = np.empty(2 * E, np.int64)
    src_global[0::2] = edges[:, 0] + atom_off[mol_of_edge]
    src_global[1::2] = edges[:, 1] + atom_off[mol_of_edge]
    tgt_global = np.empty(2 * E, np.int64)
    tgt_global[0::2] = src_global[1::2]
    tgt_global[1::2] = src_global[0::2]
    f_bonds_cat = np.empty((2 * E, ATOM_FDIM + BOND_FDIM), np.float32)
    f_bonds_cat[:, :ATOM_FDIM] = f_atoms_cat[src_global]                 # f_atoms[src] ++ f_bond, :198-199
    f_bonds_cat[0::2, ATOM_FDIM:] = fbond
    f_bonds_cat[1::2, ATOM_FDIM:] = fbond
    # incoming-bond lists in increasing bond order (a2b[a2].append(b1), a2b[a1].append(b2))
    order = np.argsort(tgt_global, kind="stable")
    bond_local = np.empty(2 * E, np.int32)
    bond_local[0::2] = 2 * e_local
    bond_local[1::2] = 2 * e_local + 1
    a2b_local = bond_local[order]
    total_atoms = int(atom_off[-1])
    counts = np.bincount(tgt_global, minlength=total_atoms).astype(np.int64)
    a2b_off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    return mol_atoms, mol_bonds, f_atoms_cat, f_bonds_cat, b2a_local, b2revb_local, a2b_off, a2b_local


def _concat_ducks(graphs):
    """Same arrays from reference-style MolGraph objects (python lists)."""
    M = len(graphs)
    mol_atoms = np.array([g.n_atoms for g in graphs], np.int32)
    mol_bonds = np.array([g.n_bonds for g in graphs], np.int32)
    fa = [np.asarray(g.f_atoms, np.float32).reshape(g.n_atoms, -1) for g in graphs]
    fb = [np.asarray(g.f_bonds, np.float32).reshape(g.n_bonds, -1) for g in graphs]
    f_atoms_cat = np.concatenate(fa, 0) if M else np.zeros((0, ATOM_FDIM), np.float32)
    f_bonds_cat = np.concatenate(fb, 0) if M else np.zeros((0, ATOM_FDIM + BOND_FDIM), np.float32)
    b2a_local = np.concatenate([np.asarray(g.b2a, np.int32) for g in graphs]) if M else np.zeros(0, np.int32)
    b2revb_local = np.concatenate([np.asarray(g.b2revb, np.int32) for g in graphs]) if M else np.zeros(0, np.int32)
    lens, vals = [], []
    for g in graphs:
        for lst in g.a2b:
            lens.append(len(lst))
            vals.extend(lst)
    a2b_off = np.concatenate([[0], np.cumsum(np.asarray(lens, np.int64))]).astype(np.int64)
    a2b_local = np.asarray(vals, np.int32)
    return mol_atoms, mol_bonds, f_atoms_cat, f_bonds_cat, b2a_local, b2revb_local, a2b_off, a2b_local


_GRAPH_KEYS = ("f_atoms", "a2b", "b2a", "b2revb", "a2a", "a_scope", "a2b_rev_t", "b2t", "a2a_t", "npad", "atom2mol",
               "b2b_t", "npad_b")


class DeviceGraph:
    """A packed batch resident in HBM: features, index arrays and backward tables.

    `f_bonds` ([nB, 84]: source-atom features ++ bond features, featurization.py:198-199) is either uploaded as is
    (host batches) or rebuilt on the device from `f_atoms[b2a]` and the bond-only columns `fbond`
    (rr_build_fbonds_f32; batches streamed from a shard file carry only the 22 bond columns)."""

    __slots__ = ("device", "nA", "nB", "K", "M", "f_atoms", "_f_bonds", "fbond", "a2b", "b2a", "b2revb", "a2a", "a_scope",
                 "a2b_rev_t", "b2t", "a2a_t", "npad", "atom2mol", "b2b_t", "npad_b", "_fb_sum", "bytes", "atom_fdim",
                 "bond_fdim")

    def __init__(self, host: dict, device):
        self.device = torch.device(device)
        self.nA, self.nB, self.K, self.M = host["nA"], host["nB"], host["K"], host["M"]
        self.atom_fdim = int(host.get("atom_fdim", ATOM_FDIM))
        self.bond_fdim = int(host.get("bond_fdim", ATOM_FDIM + BOND_FDIM)) - self.atom_fdim
        self.bytes = 0
        _bond_tables(host)
        for k in _GRAPH_KEYS + ("f_bonds",):
            t = torch.from_numpy(host[k]).to(self.device, non_blocking=False)
            setattr(self, "_f_bonds" if k == "f_bonds" else k, t)
            self.bytes += t.numel() * t.element_size()
        self.fbond = None
        self._fb_sum = None

    @classmethod
    def from_device(cls, tensors: dict, nA: int, nB: int, K: int, M: int, device, atom_fdim: int = ATOM_FDIM,
                    bond_fdim: int = BOND_FDIM) -> "DeviceGraph":
        """Wrap tensors that already live in HBM (views into a prefetched step buffer).  `tensors` holds the index
        tables, `f_atoms` and either `f_bonds` or the bond-only columns `fbond` [nB, >= bond_fdim]."""
        g = cls.__new__(cls)
        g.device = torch.device(device)
        g.nA, g.nB, g.K, g.M = int(nA), int(nB), int(K), int(M)
        g.atom_fdim, g.bond_fdim = int(atom_fdim), int(bond_fdim)
        g.bytes = 0
        for k in _GRAPH_KEYS:
            t = tensors[k]
            if not t.is_cuda:
                raise RuntimeError(f"DeviceGraph.from_device: `{k}` must already be on the GPU")
            setattr(g, k, t)
            g.bytes += t.numel() * t.element_size()
        g._f_bonds = tensors.get("f_bonds")
        g.fbond = tensors.get("fbond")
        if g._f_bonds is None and g.fbond is None:
            raise RuntimeError("DeviceGraph.from_device needs `f_bonds` or the bond-only columns `fbond`")
        g._fb_sum = None
        return g

    @property
    def f_bonds(self):
        if self._f_bonds is None:
            ld = (self.atom_fdim + self.bond_fdim + 3) // 4 * 4
            out = torch.empty(self.nB, ld, dtype=torch.float32, device=self.device)
            _lib.check(_lib.lib().rr_build_fbonds_f32(
                _lib.ptr(self.f_atoms), self.nA, self.f_atoms.stride(0), self.atom_fdim, _lib.ptr(self.b2a),
                _lib.ptr(self.fbond), self.fbond.stride(0), self.bond_fdim, self.nB, _lib.ptr(out), ld,
                _lib.stream()), "rr_build_fbonds_f32")
            self._f_bonds = out
        return self._f_bonds

    def fb_sum(self):
        """sum_k f_bonds[a2b[a,k]]  ([nA, 84]) — the bond-feature half of MPNDiff's neighbour sum
        (reference models/mpn.py:202-209).  Input-only, so it is computed once per batch and cached."""
        if self._fb_sum is None:
            from . import functions as Fn
            fb = self.f_bonds
            w = self.atom_fdim + self.bond_fdim
            buf = torch.zeros(self.nA, fb.shape[1], dtype=torch.float32, device=self.device)  # ld 84
            self._fb_sum = Fn.gather_sum(fb, self.a2b, w, out=buf)
        return self._fb_sum


class DeviceBatch:
    """A packed batch that exists in HBM only (what reactranker_amd.shards.StepPrefetcher yields): the model takes it
    wherever it takes a BatchMolGraph.  Carries the de-duplication maps of BatchMolGraph.unique() as device tensors."""

    def __init__(self, graph: DeviceGraph, unique: "DeviceBatch" = None, amap=None, amap_t=None, bmap=None, bmap_t=None):
        self.graph = graph
        self.n_mols, self.n_atoms, self.n_bonds, self.max_num_bonds = graph.M, graph.nA, graph.nB, graph.K
        self._ub = unique
        if unique is not None:
            key = str(graph.device)
            self._rr_dedup_dev = (key, amap, amap_t)             # the caches base_model.ReactionModel.forward reads
            self._rr_prefix_dev = (key, bmap, bmap_t)

    def device_graph(self, gpu=None) -> DeviceGraph:
        return self.graph

    def unique(self):
        return (self if self._ub is None else self._ub), None, None

    def unique_bonds(self):
        return None, None


def _bond_tables(host: dict) -> None:
    """Adds the bond-to-bond backward table b2b_t / npad_b (rr_derive_bond_tables) to a packed-batch dict."""
    if "b2b_t" in host:
        return
    nA, nB, K = host["nA"], host["nB"], host["K"]
    Kb = max(1, K - 1)
    host["b2b_t"] = np.empty((nB, Kb), np.int32)
    host["npad_b"] = np.empty(nB, np.float32)
    _lib.check(_lib.lib().rr_derive_bond_tables(
        _lib.np_ptr(host["a2b_rev_t"]), _lib.np_ptr(host["b2t"]), _lib.np_ptr(host["b2revb"]), _lib.np_ptr(host["b2a"]),
        _lib.np_ptr(host["npad"]), nA, nB, K, Kb, _lib.np_ptr(host["b2b_t"]), _lib.np_ptr(host["npad_b"])),
        "rr_derive_bond_tables")


def _pack(arrs, K_override: int) -> dict:
    mol_atoms, mol_bonds, f_atoms_cat, f_bonds_cat, b2a_local, b2revb_local, a2b_off, a2b_local = arrs
    L = _lib.lib()
    M = int(mol_atoms.shape[0])
    nA, nB, K = C.c_int64(), C.c_int64(), C.c_int32()
    mol_atoms = np.ascontiguousarray(mol_atoms, np.int32)
    mol_bonds = np.ascontiguousarray(mol_bonds, np.int32)
    a2b_off = np.ascontiguousarray(a2b_off, np.int64)
    _lib.check(L.rr_pack_sizes(_lib.np_ptr(mol_atoms), _lib.np_ptr(mol_bonds), M, _lib.np_ptr(a2b_off), int(K_override),
                               C.byref(nA), C.byref(nB), C.byref(K)), "rr_pack_sizes")
    nA, nB, K = int(nA.value), int(nB.value), int(K.value)
    atom_fdim = int(f_atoms_cat.shape[1]) if f_atoms_cat.ndim == 2 and f_atoms_cat.shape[1] else ATOM_FDIM
    bond_fdim = int(f_bonds_cat.shape[1]) if f_bonds_cat.ndim == 2 and f_bonds_cat.shape[1] else ATOM_FDIM + BOND_FDIM
    ld_fa = (atom_fdim + 3) // 4 * 4
    ld_fb = (bond_fdim + 3) // 4 * 4
    out = dict(
        f_atoms=np.empty((nA, ld_fa), np.float32), f_bonds=np.empty((nB, ld_fb), np.float32),
        a2b=np.empty((nA, K), np.int32), b2a=np.empty(nB, np.int32), b2revb=np.empty(nB, np.int32),
        a2a=np.empty((nA, K), np.int32), a_scope=np.empty((M, 2), np.int32),
        a2b_rev_t=np.empty((nA, K), np.int32), b2t=np.empty(nB, np.int32), a2a_t=np.empty((nA, K), np.int32),
        npad=np.empty(nA, np.float32), atom2mol=np.empty(nA, np.int32))
    f_atoms_cat = np.ascontiguousarray(f_atoms_cat, np.float32)
    f_bonds_cat = np.ascontiguousarray(f_bonds_cat, np.float32)
    b2a_local = np.ascontiguousarray(b2a_local, np.int32)
    b2revb_local = np.ascontiguousarray(b2revb_local, np.int32)
    a2b_local = np.ascontiguousarray(a2b_local, np.int32)
    _lib.check(L.rr_pack_graphs(
        _lib.np_ptr(mol_atoms), _lib.np_ptr(mol_bonds), M, _lib.np_ptr(f_atoms_cat), atom_fdim,
        _lib.np_ptr(f_bonds_cat), bond_fdim, _lib.np_ptr(b2a_local), _lib.np_ptr(b2revb_local),
        _lib.np_ptr(a2b_off), _lib.np_ptr(a2b_local), K,
        _lib.np_ptr(out["f_atoms"]), ld_fa, _lib.np_ptr(out["f_bonds"]), ld_fb,
        _lib.np_ptr(out["a2b"]), _lib.np_ptr(out["b2a"]), _lib.np_ptr(out["b2revb"]), _lib.np_ptr(out["a2a"]),
        _lib.np_ptr(out["a_scope"]), _lib.np_ptr(out["a2b_rev_t"]), _lib.np_ptr(out["b2t"]),
        _lib.np_ptr(out["a2a_t"]), _lib.np_ptr(out["npad"]), _lib.np_ptr(out["atom2mol"])), "rr_pack_graphs")
    out.update(nA=nA, nB=nB, K=K, M=M, atom_fdim=atom_fdim, bond_fdim=bond_fdim)
    _bond_tables(out)
    return out


class BatchMolGraph:
    """Drop-in for the reference BatchMolGraph (features/featurization.py:231-335)."""

    def __init__(self, mol_graphs: Sequence, atom_messages: bool = False, K: Optional[int] = None):
        if atom_messages:
            raise NotImplementedError("atom_messages=True is never used on the reference's reaction path")
        graphs = list(mol_graphs)
        self.smiles_batch = [getattr(g, "smiles", "") for g in graphs]
        self.n_mols = len(graphs)
        self.atom_fdim = get_atom_fdim()
        self.bond_fdim = get_bond_fdim() + self.atom_fdim
        specs = [g.spec if isinstance(g, MolGraph) else g for g in graphs]
        if all(isinstance(s, MolSpec) for s in specs):
            arrs = _concat_specs(specs)
        else:
            arrs = _concat_ducks(specs)
        self._init_from_arrays(arrs, 0 if K is None else int(K), self.smiles_batch, specs)

    def _init_from_arrays(self, arrs, K: int, smiles, specs):
        self.smiles_batch = list(smiles)
        self.n_mols = len(self.smiles_batch)
        self.atom_fdim = get_atom_fdim()
        self.bond_fdim = get_bond_fdim() + self.atom_fdim
        self._specs = specs                      # kept for unique(): repeated molecule objects are de-duplicated by identity
        self._mol_ids = None if specs is not None else np.arange(self.n_mols, dtype=np.int32)
        self._host = _pack(arrs, int(K))
        h = self._host
        self.n_atoms, self.n_bonds = h["nA"], h["nB"]
        self.max_num_bonds = h["K"]
        self.a_scope: List[Tuple[int, int]] = [tuple(int(v) for v in r) for r in h["a_scope"]]
        boff = np.concatenate([[1], 1 + np.cumsum(arrs[1], dtype=np.int64)])
        self.b_scope: List[Tuple[int, int]] = [(int(boff[i]), int(arrs[1][i])) for i in range(self.n_mols)]
        self.b2b = None
        self._dev = {}

    # ---- reference-typed views (FloatTensor / LongTensor on the host) -------------------------------
    @property
    def f_atoms(self):
        return torch.from_numpy(np.ascontiguousarray(self._host["f_atoms"][:, :self._host["atom_fdim"]]))

    @property
    def f_bonds(self):
        return torch.from_numpy(np.ascontiguousarray(self._host["f_bonds"][:, :self._host["bond_fdim"]]))

    @property
    def a2b(self):
        return torch.from_numpy(self._host["a2b"].astype(np.int64))

    @property
    def b2a(self):
        return torch.from_numpy(self._host["b2a"].astype(np.int64))

    @property
    def b2revb(self):
        return torch.from_numpy(self._host["b2revb"].astype(np.int64))

    @property
    def a2a(self):
        return torch.from_numpy(self._host["a2a"].astype(np.int64))

    def get_components(self):
        return self.f_atoms, self.f_bonds, self.a2b, self.b2a, self.b2revb, self.a_scope, self.b_scope

    def unique_bonds(self):
        """(bmap [nB], bmap_t [nB_unique, C]) — bond-row maps of unique(); see there."""
        self.unique()
        return self._unique_bonds

    def unique(self):
        """De-duplicated view of the batch: (BatchMolGraph of the distinct molecule objects, atom map).

        Every candidate of a query repeats the SAME reactant object (reference train_listwise.py:188 ->
        load_reactions.py:574-577), so encoder(r) does C-fold redundant work.
        `unique()` packs each distinct object once (same pad width K, so hazard H1 is unaffected) and returns
        `amap` [nA] int32 (full-batch atom row -> row in the unique batch, row 0 -> 0) and its transpose
        `amap_t` [nA_unique, C] (-1 padded) for the backward segment sum.  Identity-based: two equal molecules
        built as different objects are not merged."""
        if getattr(self, "_unique", None) is not None:
            return self._unique
        uidx = self.molecule_ids()
        n_u = int(uidx.max()) + 1 if len(uidx) else 0
        firsts = np.full(n_u, -1, np.int64)
        for m in range(len(uidx) - 1, -1, -1):
            firsts[uidx[m]] = m                                # first occurrence of every distinct molecule
        ub = self.subset(firsts)
        if ub.max_num_bonds != self.max_num_bonds:
            raise RuntimeError("unique(): pad width mismatch")
        h, hu = self._host, ub._host
        nA = h["nA"]
        amap = np.zeros(nA, np.int32)
        uidx = np.asarray(uidx, np.int64)
        if len(uidx):
            start, size = h["a_scope"][:, 0].astype(np.int64), h["a_scope"][:, 1].astype(np.int64)
            ustart = hu["a_scope"][uidx, 0].astype(np.int64)
            rep = np.repeat(np.arange(len(uidx)), size)
            within = np.arange(int(size.sum())) - np.repeat(np.cumsum(size) - size, size)
            amap[start[rep] + within] = (ustart[rep] + within).astype(np.int32)
        def transpose(mp, n_u):
            counts = np.bincount(mp, minlength=n_u)
            cmax = int(max(1, counts.max()))
            tt = np.full((n_u, cmax), -1, np.int32)
            order = np.argsort(mp, kind="stable")
            pos = np.arange(mp.shape[0]) - np.repeat(np.cumsum(counts) - counts, counts)
            tt[mp[order], pos] = order.astype(np.int32)
            return tt
        amap_t = transpose(amap, hu["nA"])
        # the same maps for directed bonds (shared-prefix path of the reactant encoder)
        bmap = np.zeros(h["nB"], np.int32)
        if len(uidx):
            bsc = np.asarray(self.b_scope, np.int64).reshape(-1, 2)
            bsu = np.asarray(ub.b_scope, np.int64).reshape(-1, 2)
            bstart, bsize = bsc[:, 0], bsc[:, 1]
            rep = np.repeat(np.arange(len(uidx)), bsize)
            within = np.arange(int(bsize.sum())) - np.repeat(np.cumsum(bsize) - bsize, bsize)
            bmap[bstart[rep] + within] = (bsu[uidx, 0][rep] + within).astype(np.int32)
        self._unique_bonds = (bmap, transpose(bmap, hu["nB"]))
        self._unique = (ub, amap, amap_t)
        return self._unique

    def molecule_ids(self) -> np.ndarray:
        """[n_mols] int32: id of the distinct molecule each batch entry repeats (ids in order of first occurrence).
        Identity-based when the batch was built from molecule objects; stored with packed batches (save_packed)."""
        ids = getattr(self, "_mol_ids", None)
        if ids is None:
            first, out = {}, []
            for s in self._specs:
                k = id(s)
                if k not in first:
                    first[k] = len(first)
                out.append(first[k])
            ids = np.asarray(out, np.int32)
            self._mol_ids = ids
        return ids

    def subset(self, mol_ids) -> "BatchMolGraph":
        """A new batch holding the molecules `mol_ids` (in that order), rebuilt from this batch's packed arrays with
        the same pad width - e.g. the distinct reactants of unique(), or a rank's block of whole queries."""
        mol_ids = np.asarray(mol_ids, np.int64).reshape(-1)
        h = self._host
        K = h["K"]
        afd, bfd = h["atom_fdim"], h["bond_fdim"]
        a_sc = h["a_scope"].astype(np.int64)
        b_sc = np.asarray(self.b_scope, np.int64).reshape(-1, 2)
        mol_atoms = a_sc[mol_ids, 1].astype(np.int32) if len(mol_ids) else np.zeros(0, np.int32)
        mol_bonds = b_sc[mol_ids, 1].astype(np.int32) if len(mol_ids) else np.zeros(0, np.int32)

        def rows(scope_start, sizes):
            rep = np.repeat(np.arange(len(sizes)), sizes)
            within = np.arange(int(sizes.sum())) - np.repeat(np.cumsum(sizes) - sizes, sizes)
            return scope_start[rep] + within, rep
        arow, amol = rows(a_sc[mol_ids, 0], mol_atoms.astype(np.int64)) if len(mol_ids) else (np.zeros(0, np.int64),) * 2
        brow, bmol = rows(b_sc[mol_ids, 0], mol_bonds.astype(np.int64)) if len(mol_ids) else (np.zeros(0, np.int64),) * 2
        f_atoms_cat = np.ascontiguousarray(h["f_atoms"][arow, :afd])
        f_bonds_cat = np.ascontiguousarray(h["f_bonds"][brow, :bfd])
        b2a_local = (h["b2a"][brow] - a_sc[mol_ids, 0][bmol]).astype(np.int32)
        b2revb_local = (h["b2revb"][brow] - b_sc[mol_ids, 0][bmol]).astype(np.int32)
        deg = (K - h["npad"][arow]).astype(np.int64)
        a2b_off = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
        a2b_rows = h["a2b"][arow]                                            # [n, K], real entries first
        keep = np.arange(K)[None, :] < deg[:, None]
        a2b_local = (a2b_rows - b_sc[mol_ids, 0][amol][:, None])[keep].astype(np.int32)
        sub = BatchMolGraph.__new__(BatchMolGraph)
        sub._init_from_arrays((mol_atoms, mol_bonds, f_atoms_cat, f_bonds_cat, b2a_local, b2revb_local, a2b_off, a2b_local),
                              K, [self.smiles_batch[i] for i in mol_ids], None)
        return sub

    def get_a2a(self):
        return self.a2a

    def get_smiles(self):
        return self.smiles_batch

    # ---- HBM-resident form --------------------------------------------------------------------------
    def device_graph(self, gpu) -> DeviceGraph:
        dev = torch.device("cuda", torch.cuda.current_device() if gpu is None else int(gpu)) \
            if not isinstance(gpu, torch.device) else gpu
        key = str(dev)
        if key not in self._dev:
            self._dev[key] = DeviceGraph(self._host, dev)
        return self._dev[key]


def device_graph_of(batch, gpu) -> DeviceGraph:
    """DeviceGraph for our BatchMolGraph, or for any object exposing the reference's
    get_components()/get_a2a() contract (e.g. the reference's own BatchMolGraph)."""
    if isinstance(batch, DeviceGraph):
        return batch
    if isinstance(batch, (BatchMolGraph, DeviceBatch)):
        return batch.device_graph(gpu)
    cache = getattr(batch, "_rr_device_graphs", None)
    if cache is None:
        cache = {}
        try:
            batch._rr_device_graphs = cache
        except AttributeError:
            pass
    dev = torch.device("cuda", torch.cuda.current_device() if gpu is None else int(gpu))
    key = str(dev)
    if key in cache:
        return cache[key]
    f_atoms, f_bonds, a2b, b2a, b2revb, a_scope, _ = batch.get_components()
    fa = np.asarray(f_atoms, np.float32)
    fb = np.asarray(f_bonds, np.float32)
    a2b_np = np.ascontiguousarray(np.asarray(a2b), np.int32)
    b2a_np = np.ascontiguousarray(np.asarray(b2a), np.int32)
    b2revb_np = np.ascontiguousarray(np.asarray(b2revb), np.int32)
    nA, K = a2b_np.shape
    nB = b2a_np.shape[0]
    scope = np.ascontiguousarray(np.asarray(a_scope, np.int32).reshape(-1, 2))
    M = scope.shape[0]
    ld_fa, ld_fb = (fa.shape[1] + 3) // 4 * 4, (fb.shape[1] + 3) // 4 * 4
    host = dict(f_atoms=np.zeros((nA, ld_fa), np.float32), f_bonds=np.zeros((nB, ld_fb), np.float32),
                a2b=a2b_np, b2a=b2a_np, b2revb=b2revb_np, a2a=np.empty((nA, K), np.int32), a_scope=scope,
                a2b_rev_t=np.empty((nA, K), np.int32), b2t=np.empty(nB, np.int32), a2a_t=np.empty((nA, K), np.int32),
                npad=np.empty(nA, np.float32), atom2mol=np.empty(nA, np.int32), nA=nA, nB=nB, K=K, M=M)
    host["f_atoms"][:, :fa.shape[1]] = fa
    host["f_bonds"][:, :fb.shape[1]] = fb
    _lib.check(_lib.lib().rr_derive_tables(
        _lib.np_ptr(a2b_np), _lib.np_ptr(b2a_np), _lib.np_ptr(b2revb_np), nA, nB, K, _lib.np_ptr(scope), M,
        _lib.np_ptr(host["a2a"]), _lib.np_ptr(host["a2b_rev_t"]), _lib.np_ptr(host["b2t"]),
        _lib.np_ptr(host["a2a_t"]), _lib.np_ptr(host["npad"]), _lib.np_ptr(host["atom2mol"])), "rr_derive_tables")
    dg = DeviceGraph(host, dev)
    cache[key] = dg
    return dg


def mol2graph(smiles_batch, reaction: bool = True):
    """Reference featurization.py:338-350: SMILES list -> BatchMolGraph (needs RDKit, see MolGraph)."""
    return BatchMolGraph([MolGraph(s, reaction=reaction) for s in smiles_batch])


# ---- packed on-disk format: the packer's host arrays as one .npz per batch ------------
_PACK_KEYS = ("f_atoms", "f_bonds", "a2b", "b2a", "b2revb", "a2a", "a_scope", "a2b_rev_t", "b2t", "a2a_t", "npad",
              "atom2mol")


def save_packed(batch: BatchMolGraph, path: str) -> None:
    """Write a packed batch (features, index arrays, backward tables, molecule identities) so a dataset is packed
    once, not per epoch (the reference re-runs BatchMolGraph's Python list packing for every batch of every epoch).
    For whole training steps streamed from disk see reactranker_amd.shards."""
    h = batch._host
    np.savez(path, **{k: h[k] for k in _PACK_KEYS},
             meta=np.asarray([h["nA"], h["nB"], h["K"], h["M"], h["atom_fdim"], h["bond_fdim"]], np.int64),
             b_scope=np.asarray(batch.b_scope, np.int64).reshape(-1, 2), mol_ids=batch.molecule_ids())


def load_packed(path: str) -> BatchMolGraph:
    """Inverse of save_packed: a BatchMolGraph backed by the stored arrays (no re-packing).  The stored molecule
    identities let unique() / unique_bonds() (reactant de-duplication) work exactly as on the original batch."""
    d = np.load(path if path.endswith(".npz") else path + ".npz")
    b = BatchMolGraph.__new__(BatchMolGraph)
    nA, nB, K, M, afd, bfd = (int(v) for v in d["meta"])
    b._host = {k: np.ascontiguousarray(d[k]) for k in _PACK_KEYS}
    b._host.update(nA=nA, nB=nB, K=K, M=M, atom_fdim=afd, bond_fdim=bfd)
    _bond_tables(b._host)
    b.smiles_batch, b.n_mols = [""] * M, M
    b.atom_fdim, b.bond_fdim = get_atom_fdim(), get_bond_fdim() + get_atom_fdim()
    b.n_atoms, b.n_bonds, b.max_num_bonds = nA, nB, K
    b.a_scope = [tuple(int(v) for v in r) for r in b._host["a_scope"]]
    b.b_scope = [tuple(int(v) for v in r) for r in d["b_scope"]]
    b.b2b, b._dev, b._specs = None, {}, None
    b._mol_ids = np.asarray(d["mol_ids"], np.int32) if "mol_ids" in d.files else np.arange(M, dtype=np.int32)
    return b
