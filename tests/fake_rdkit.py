"""A stand-in for the few RDKit names the featuriser touches, serving molecules from plain descriptions.

RDKit is not installed in the build container, and nothing in the reference pins what RDKit returns for a SMILES.  What
CAN be pinned is everything on this side of RDKit: one-hot layouts, the unknown slots, atom order by map number, bond
numbering.  tools/make_golden.py serves these descriptions to the REFERENCE's own MolGraph through this namespace and
stores what it produced (tests/golden/featurizer.npz); tests/test_host_cpu.py serves the same descriptions to
reactranker_amd.rdkit_features.  Test infrastructure only.
"""
import types

import numpy as np


class HybridizationType:
    UNSPECIFIED, SP, SP2, SP3, SP3D, SP3D2, OTHER = range(7)


class BondType:
    UNSPECIFIED, SINGLE, DOUBLE, TRIPLE, AROMATIC = range(5)


class Atom:
    def __init__(self, idx, d):
        self.idx, self.d = idx, d

    def GetIdx(self): return self.idx
    def GetSymbol(self): return self.d["symbol"]
    def GetTotalDegree(self): return self.d["degree"]
    def GetFormalCharge(self): return self.d["charge"]
    def GetChiralTag(self): return self.d["chiral"]
    def GetTotalNumHs(self): return self.d["hs"]
    def GetNumRadicalElectrons(self): return self.d["radicals"]
    def GetHybridization(self): return self.d["hyb"]
    def GetIsAromatic(self): return bool(self.d["aromatic"])
    def GetMass(self): return self.d["mass"]
    def IsInRingSize(self, n): return n in self.d["rings"]
    def GetAtomMapNum(self): return self.d["map"]


class Bond:
    def __init__(self, d):
        self.d = d

    def GetBeginAtomIdx(self): return self.d["a"]
    def GetEndAtomIdx(self): return self.d["b"]
    def GetBondType(self): return self.d["type"]
    def GetIsConjugated(self): return bool(self.d["conj"])
    def IsInRing(self): return len(self.d["rings"]) > 0
    def IsInRingSize(self, n): return n in self.d["rings"]
    def GetStereo(self): return self.d["stereo"]


class Mol:
    def __init__(self, desc):
        self.atoms = [Atom(i, a) for i, a in enumerate(desc["atoms"])]
        self.bonds = [Bond(b) for b in desc["bonds"]]
        self.explicit_h = False

    def GetNumAtoms(self): return len(self.atoms)
    def GetAtoms(self): return list(self.atoms)
    def GetBonds(self): return list(self.bonds)

    def GetBondBetweenAtoms(self, i, j):
        for b in self.bonds:
            if {b.d["a"], b.d["b"]} == {i, j}:
                return b
        return None


def descriptions(seed: int = 7):
    """name -> description.  Covers every slot of every one-hot block including the unknown ones, repeated and zero
    atom-map numbers (stable order), a bond of unknown type and one with type None, an atom without bonds."""
    rng = np.random.default_rng(seed)
    symbols = ["H", "C", "N", "O", "S", "F", "Si", "P", "Cl", "Br", "Mg", "Na", "I", "B", "K", "Zn"]
    out = {}
    for m, n in enumerate((1, 2, 5, 9, 17, 23)):
        atoms = []
        for i in range(n):
            atoms.append(dict(symbol=symbols[(i + 3 * m) % len(symbols)], degree=int(rng.integers(0, 7)),
                              charge=int(rng.integers(-3, 4)), chiral=int(rng.integers(0, 5)), hs=int(rng.integers(0, 6)),
                              radicals=int(rng.integers(0, 6)), hyb=int(rng.integers(0, 7)), aromatic=int(rng.integers(0, 2)),
                              mass=float(np.round(rng.uniform(1.0, 130.0), 3)),
                              rings=sorted(int(r) for r in rng.choice(np.arange(3, 12), size=int(rng.integers(0, 3)), replace=False)),
                              map=int(rng.integers(0, max(2, n // 2)))))            # repeats and zeros on purpose
        pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
        rng.shuffle(pairs)
        bonds = []
        for k, (i, j) in enumerate(pairs[:min(len(pairs), n + 2)]):
            if n > 3 and (i == n - 1 or j == n - 1):
                continue                                                            # the last atom stays without bonds
            a, b = (i, j) if rng.integers(0, 2) else (j, i)                         # begin / end in either order
            t = [BondType.SINGLE, BondType.DOUBLE, BondType.TRIPLE, BondType.AROMATIC, BondType.UNSPECIFIED, None][k % 6]
            bonds.append(dict(a=a, b=b, type=t, conj=int(rng.integers(0, 2)),
                              rings=sorted(int(r) for r in rng.choice(np.arange(3, 12), size=int(rng.integers(0, 3)), replace=False)),
                              stereo=int(rng.integers(0, 8))))
        out[f"mol{m}_{n}atoms"] = dict(atoms=atoms, bonds=bonds)
    return out


def chem_namespace(descs):
    """An object with the attributes of `rdkit.Chem` that featurisation code uses; MolFromSmiles looks the 'SMILES' up
    in `descs`."""
    chem = types.SimpleNamespace()

    class SmilesParserParams:
        removeHs = False

    def mol_from_smiles(s, params=None):
        return Mol(descs[s]) if s in descs else None

    def add_hs(mol):
        mol.explicit_h = True
        return mol

    chem.SmilesParserParams = SmilesParserParams
    chem.MolFromSmiles = mol_from_smiles
    chem.MolFromInchi = lambda s, removeHs=False: None
    chem.AddHs = add_hs
    chem.RemoveHs = lambda mol: mol
    chem.BondType = BondType
    chem.rdchem = types.SimpleNamespace(HybridizationType=HybridizationType, Atom=Atom, Bond=Bond)
    chem.Mol = Mol
    return chem
