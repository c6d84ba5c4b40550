"""Seeded synthetic reaction-graph generator.

RDKit is not available where this runs, so real SMILES cannot be featurised.  The
generator emits molecule-like graphs that respect the reference feature layout:

* atom features: 61 floats = one-hot blocks 16/6/6/5/6/6/6 + aromatic bit + mass*0.01
  + 8 ring-size flags            (reference: features/featurization.py:45-63, 76-96)
* bond features: 22 floats = null tag, 4 bond types, conjugated, in-ring, 8 ring sizes,
  7-way stereo one-hot           (reference: features/featurization.py:103-132)
* directed-bond numbering exactly as MolGraph builds it: for a1 < a2 bonded,
  b1 = a1->a2, b2 = a2->a1, f_bonds[b] = f_atoms[src] ++ f_bond
                                 (reference: features/featurization.py:184-210)

A query is one reactant graph plus `n_cand` candidate products; every product has the
same atoms in the same order (the model subtracts atom hiddens row by row,
reference: models/base_model.py:168) and differs by one moved bond.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

ATOM_FDIM = 61
BOND_FDIM = 22
_ONEHOT_BLOCKS = (16, 6, 6, 5, 6, 6, 6)
_MASSES = np.array([1.008, 12.011, 14.007, 15.999, 32.06, 18.998, 28.085, 30.974,
                    35.45, 79.904, 24.305, 22.990, 126.90, 10.81, 39.098, 50.0], dtype=np.float64)
_SYMBOL_P = np.array([0.45, 0.35, 0.06, 0.08, 0.01, 0.01, 0.005, 0.005, 0.01, 0.005,
                      0.002, 0.002, 0.002, 0.002, 0.002, 0.005])
_SYMBOL_P = _SYMBOL_P / _SYMBOL_P.sum()


@dataclass
class MolSpec:
    """One molecule in compact array form (what a reference MolGraph holds as lists)."""
    n_atoms: int
    f_atoms: np.ndarray      # [n_atoms, 61] float32
    edges: np.ndarray        # [E, 2] int32, a1 < a2, lexicographically sorted
    f_bond: np.ndarray       # [E, 22] float32
    smiles: str = "synthetic"

    @property
    def n_bonds(self) -> int:
        return 2 * int(self.edges.shape[0])

    def directed(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray, List[List[int]]]:
        """f_bonds [2E,83], b2a [2E], b2revb [2E], a2b (list per atom) in reference numbering."""
        E = self.edges.shape[0]
        a1 = self.edges[:, 0]
        a2 = self.edges[:, 1]
        b2a = np.empty(2 * E, dtype=np.int32)
        b2a[0::2] = a1
        b2a[1::2] = a2
        b2revb = np.empty(2 * E, dtype=np.int32)
        b2revb[0::2] = np.arange(E) * 2 + 1
        b2revb[1::2] = np.arange(E) * 2
        f_bonds = np.empty((2 * E, ATOM_FDIM + BOND_FDIM), dtype=np.float32)
        f_bonds[:, :ATOM_FDIM] = self.f_atoms[b2a]
        f_bonds[0::2, ATOM_FDIM:] = self.f_bond
        f_bonds[1::2, ATOM_FDIM:] = self.f_bond
        a2b: List[List[int]] = [[] for _ in range(self.n_atoms)]
        for e in range(E):
            a2b[int(a2[e])].append(2 * e)        # b1 = a1->a2 is incoming to a2
            a2b[int(a1[e])].append(2 * e + 1)    # b2 = a2->a1 is incoming to a1
        return f_bonds, b2a, b2revb, a2b


class ListMolGraph:
    """Duck-typed twin of the reference MolGraph (python lists), built from a MolSpec.

    Only used to drive the *reference's own* BatchMolGraph when generating golden vectors
    and in the faithful CPU baseline; the product path never touches python lists.
    """

    def __init__(self, spec: MolSpec):
        f_bonds, b2a, b2revb, a2b = spec.directed()
        self.smiles = spec.smiles
        self.n_atoms = spec.n_atoms
        self.n_bonds = spec.n_bonds
        self.f_atoms = spec.f_atoms.tolist()
        self.f_bonds = f_bonds.tolist()
        self.a2b = a2b
        self.b2a = b2a.tolist()
        self.b2revb = b2revb.tolist()


def _atom_features(rng: np.random.Generator, n: int) -> np.ndarray:
    f = np.zeros((n, ATOM_FDIM), dtype=np.float32)
    col = 0
    sym = rng.choice(16, size=n, p=_SYMBOL_P)
    for bi, width in enumerate(_ONEHOT_BLOCKS):
        if bi == 0:
            pick = sym
        else:
            # skewed towards the low categories, like real molecules
            pick = np.minimum(rng.geometric(0.55, size=n) - 1, width - 1)
        f[np.arange(n), col + pick] = 1.0
        col += width
    f[:, col] = (rng.random(n) < 0.2).astype(np.float32)           # aromatic
    f[:, col + 1] = (_MASSES[sym] * 0.01).astype(np.float32)       # mass * 0.01
    f[:, col + 2:col + 10] = (rng.random((n, 8)) < 0.06).astype(np.float32)  # ring sizes 3..10
    return f


def _bond_features(rng: np.random.Generator, e: int) -> np.ndarray:
    f = np.zeros((e, BOND_FDIM), dtype=np.float32)
    if e == 0:
        return f
    bt = rng.choice(4, size=e, p=[0.75, 0.15, 0.03, 0.07])
    f[np.arange(e), 1 + bt] = 1.0
    f[:, 5] = (rng.random(e) < 0.2).astype(np.float32)             # conjugated
    ring = rng.random(e) < 0.15
    f[:, 6] = ring.astype(np.float32)
    rs = rng.integers(0, 8, size=e)
    f[np.arange(e)[ring], 7 + rs[ring]] = 1.0
    st = np.minimum(rng.geometric(0.8, size=e) - 1, 6)
    f[np.arange(e), 15 + st] = 1.0
    return f


def _sorted_edges(edge_set) -> np.ndarray:
    if not edge_set:
        return np.zeros((0, 2), dtype=np.int32)
    return np.array(sorted(edge_set), dtype=np.int32)


def random_reactant(rng: np.random.Generator, n_atoms: int, max_degree: int = 4) -> MolSpec:
    deg = np.zeros(n_atoms, dtype=np.int64)
    edges = set()
    for i in range(1, n_atoms):
        cand = np.flatnonzero(deg[:i] < max_degree)
        j = int(cand[rng.integers(0, len(cand))])
        edges.add((j, i))
        deg[i] += 1
        deg[j] += 1
    if n_atoms >= 4 and rng.random() < 0.5:      # 0-1 ring closure
        for _ in range(8):
            a, b = sorted(int(x) for x in rng.choice(n_atoms, size=2, replace=False))
            if (a, b) not in edges and deg[a] < max_degree and deg[b] < max_degree:
                edges.add((a, b))
                deg[a] += 1
                deg[b] += 1
                break
    e = _sorted_edges(edges)
    return MolSpec(n_atoms, _atom_features(rng, n_atoms), e, _bond_features(rng, len(e)), "R")


def random_product(rng: np.random.Generator, reactant: MolSpec, max_degree: int = 4) -> MolSpec:
    """Reactant with one bond removed and one formed elsewhere (same atoms, same order)."""
    n = reactant.n_atoms
    edge_list = [tuple(int(v) for v in e) for e in reactant.edges]
    feat = {e: reactant.f_bond[i] for i, e in enumerate(edge_list)}
    edges = set(edge_list)
    if edge_list:
        victim = edge_list[int(rng.integers(0, len(edge_list)))]
        edges.discard(victim)
        feat.pop(victim)
    deg = np.zeros(n, dtype=np.int64)
    for a, b in edges:
        deg[a] += 1
        deg[b] += 1
    if n >= 2:
        for _ in range(16):
            a, b = sorted(int(x) for x in rng.choice(n, size=2, replace=False))
            if (a, b) not in edges and deg[a] < max_degree and deg[b] < max_degree:
                edges.add((a, b))
                feat[(a, b)] = _bond_features(rng, 1)[0]
                break
    e = _sorted_edges(edges)
    fb = np.stack([feat[tuple(int(v) for v in row)] for row in e]) if len(e) else \
        np.zeros((0, BOND_FDIM), dtype=np.float32)
    return MolSpec(n, reactant.f_atoms, e, fb.astype(np.float32), "P")


@dataclass
class QueryBatch:
    """A step's worth of whole queries: per-candidate reactant/product specs + targets."""
    r_specs: List[MolSpec]
    p_specs: List[MolSpec]
    scope: List[int]                 # candidates per query
    targets: np.ndarray              # [M] float32, distinct within a query
    add_features: np.ndarray         # [M, 1] float32 ("temp")


def make_queries(seed: int, n_queries: int, n_cand, atoms_lo: int = 10, atoms_hi: int = 24) -> QueryBatch:
    """`n_cand` is an int (equal lists) or a sequence of per-query list lengths (ragged)."""
    rng = np.random.default_rng(seed)
    if np.isscalar(n_cand):
        scope = [int(n_cand)] * n_queries
    else:
        scope = [int(c) for c in n_cand]
        assert len(scope) == n_queries
    r_specs: List[MolSpec] = []
    p_specs: List[MolSpec] = []
    targets = []
    for q in range(n_queries):
        n_atoms = int(rng.integers(atoms_lo, atoms_hi + 1))
        r = random_reactant(rng, n_atoms)
        for _ in range(scope[q]):
            r_specs.append(r)
            p_specs.append(random_product(rng, r))
        t = rng.standard_normal(scope[q]).astype(np.float32)
        # distinct targets inside a query (argsort in the reference is unstable on ties)
        while len(np.unique(t)) != len(t):
            t = rng.standard_normal(scope[q]).astype(np.float32)
        targets.append(t)
    m = sum(scope)
    add = rng.random((m, 1)).astype(np.float32)
    return QueryBatch(r_specs, p_specs, scope,
                      np.concatenate(targets) if targets else np.zeros(0, np.float32), add)


def seeded_weights(shapes: dict, seed: int, scale: float = None) -> dict:
    """Reproducible weights without torch's RNG: name -> float32 array.

    uniform(-a, a) with a = 1/sqrt(fan_in) (a = `scale` if given); 1-D tensors use the
    fan_in of the matching weight so biases are non-trivial (hazard H1 needs b != 0).
    """
    rng = np.random.default_rng(seed)
    out = {}
    for name in sorted(shapes):
        shp = tuple(shapes[name])
        if name.endswith("cached_zero_vector"):
            out[name] = np.zeros(shp, dtype=np.float32)
            continue
        if len(shp) == 2:
            a = scale if scale is not None else 1.0 / np.sqrt(shp[1])
        else:
            wname = name[:-len("bias")] + "weight"
            fan_in = shapes[wname][1] if wname in shapes else shp[0]
            a = scale if scale is not None else 1.0 / np.sqrt(fan_in)
        out[name] = rng.uniform(-a, a, size=shp).astype(np.float32)
    return out
