"""Optional SMILES -> MolSpec featuriser ("optional RDKit featuriser when the package exists").

Mirrors the feature LAYOUT of the reference's reactranker/features/featurization.py - atom_features (:68-100, 61
numbers), bond_features (:103-132, 22 numbers) and MolGraph.__init__'s atom order and bond numbering (:148-210) - and
hands the result to the packer as a `synth.MolSpec` (arrays; no per-bond Python lists downstream).  RDKit does the
chemistry; it is imported lazily, so everything else in the package works without it, and `MolGraph(smiles)` says
what is missing when it is not installed.

The layout is pinned without RDKit: tests/golden/featurizer.npz holds what the reference's own MolGraph produced for
a handful of molecule descriptions served through a stand-in `Chem` namespace (tools/make_golden.py gen_featurizer),
and tests/test_host_cpu.py feeds the same descriptions through this module.  What RDKit itself returns for a SMILES is
RDKit's business and is not pinned by anything in the reference either.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from .synth import ATOM_FDIM, BOND_FDIM, MolSpec

SYMBOLS = ("H", "C", "N", "O", "S", "F", "Si", "P", "Cl", "Br", "Mg", "Na", "I", "B", "K")   # featurization.py:45
RING_SIZES = tuple(range(3, 11))                                                            # :88-97, :123-130


def _chem():
    try:
        from rdkit import Chem
    except ImportError as e:                                        # the one place that needs the package
        raise RuntimeError("SMILES featurisation needs RDKit (pip install rdkit); without it pass MolSpec arrays "
                           "(MolGraph.from_spec) or objects with the reference MolGraph's attributes") from e
    return Chem


def _one_hot(out: List[float], value, choices: Sequence) -> None:
    """len(choices) + 1 slots; a value outside `choices` lights the last one (onek_encoding_unk, :29-42)."""
    slot = len(choices)
    for i, c in enumerate(choices):
        if c == value:
            slot = i
            break
    out.extend(1.0 if i == slot else 0.0 for i in range(len(choices) + 1))


def atom_features(atom, chem=None) -> List[float]:
    """61 numbers: symbol(16) degree(6) charge(6) chirality(5) Hs(6) radicals(6) hybridisation(6) aromatic mass/100
    ring-size 3..10 membership (featurization.py:47-63 blocks, :76-97 order)."""
    chem = chem or _chem()
    hyb = chem.rdchem.HybridizationType
    f: List[float] = []
    _one_hot(f, atom.GetSymbol(), SYMBOLS)
    _one_hot(f, atom.GetTotalDegree(), (0, 1, 2, 3, 4))
    _one_hot(f, atom.GetFormalCharge(), (-2, -1, 0, 1, 2))
    _one_hot(f, atom.GetChiralTag(), (0, 1, 2, 3))
    _one_hot(f, atom.GetTotalNumHs(), (0, 1, 2, 3, 4))
    _one_hot(f, atom.GetNumRadicalElectrons(), (0, 1, 2, 3, 4))
    _one_hot(f, atom.GetHybridization(), (hyb.SP, hyb.SP2, hyb.SP3, hyb.SP3D, hyb.SP3D2))
    f.append(1.0 if atom.GetIsAromatic() else 0.0)
    f.append(atom.GetMass() * 0.01)
    f.extend(1.0 if atom.IsInRingSize(n) else 0.0 for n in RING_SIZES)
    return f


def bond_features(bond, chem=None) -> List[float]:
    """22 numbers: a 'no bond' tag, single / double / triple / aromatic, conjugated, in a ring, ring-size 3..10
    membership, stereo one-hot over 0..5 (+1 unknown slot) (featurization.py:110-132)."""
    if bond is None:
        return [1.0] + [0.0] * (BOND_FDIM - 1)
    chem = chem or _chem()
    bt = bond.GetBondType()
    kinds = chem.BondType
    f = [0.0]
    f.extend(1.0 if bt == k else 0.0 for k in (kinds.SINGLE, kinds.DOUBLE, kinds.TRIPLE, kinds.AROMATIC))
    known = bt is not None
    f.append(1.0 if known and bond.GetIsConjugated() else 0.0)
    f.append(1.0 if known and bond.IsInRing() else 0.0)
    f.extend(1.0 if known and bond.IsInRingSize(n) else 0.0 for n in RING_SIZES)
    _one_hot(f, int(bond.GetStereo()), tuple(range(6)))
    return f


def spec_from_mol(mol, smiles: str = "", reaction: bool = True, chem=None) -> MolSpec:
    """Arrays of one molecule.  reaction=True orders the atoms by atom-map number (stable, like sorted(); the
    reference needs reactant and product atoms aligned for p_h - r_h, featurization.py:159-168); bonds are numbered
    over atom pairs (a1 < a2) in that order (:178-208), which is MolSpec's edge order."""
    chem = chem or _chem()
    atoms = list(mol.GetAtoms())
    if reaction:
        atoms.sort(key=lambda a: a.GetAtomMapNum())
    n = len(atoms)
    f_atoms = np.asarray([atom_features(a, chem) for a in atoms], np.float32).reshape(n, ATOM_FDIM)
    pos = {a.GetIdx(): i for i, a in enumerate(atoms)}
    found = []
    for b in mol.GetBonds():                                        # E bonds instead of the reference's n^2 / 2 pair probes
        i, j = pos[b.GetBeginAtomIdx()], pos[b.GetEndAtomIdx()]
        found.append(((i, j) if i < j else (j, i), b))
    found.sort(key=lambda t: t[0])
    edges = np.asarray([e for e, _ in found], np.int32).reshape(len(found), 2)
    f_bond = np.asarray([bond_features(b, chem) for _, b in found], np.float32).reshape(len(found), BOND_FDIM)
    return MolSpec(n_atoms=n, f_atoms=f_atoms, edges=edges, f_bond=f_bond, smiles=smiles)


def str_to_mol(string: str, explicit_hydrogens: bool = True, chem=None):
    """InChI or SMILES -> RDKit molecule with explicit hydrogens kept / added (featurization.py:8-26)."""
    chem = chem or _chem()
    if string.startswith("InChI"):
        mol = chem.MolFromInchi(string, removeHs=not explicit_hydrogens)
    else:
        params = chem.SmilesParserParams()
        params.removeHs = not explicit_hydrogens
        mol = chem.MolFromSmiles(string, params)
    if mol is None:
        raise ValueError(f"RDKit could not parse {string!r}")
    return chem.AddHs(mol) if explicit_hydrogens else chem.RemoveHs(mol)


def spec_from_smiles(smiles: str, reaction: bool = True, chem=None) -> MolSpec:
    chem = chem or _chem()
    return spec_from_mol(str_to_mol(smiles, True, chem), smiles, reaction, chem)


def available() -> bool:
    try:
        _chem()
        return True
    except RuntimeError:
        return False
