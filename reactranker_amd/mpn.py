"""MPN / MPNDiff — host-side mirror of reference `reactranker/models/mpn.py`.

Same constructor arguments, parameter names and forward signatures as the reference
(MPN :14-21,61-64; MPNDiff :130-136,170-174); the math runs in the HIP kernels behind
include/reactranker_hip.h through reactranker_amd.functions.
"""
from typing import List

import numpy as np
import torch
import torch.nn as nn

from . import functions as Fn
from .featurization import device_graph_of


def _fresh_seed() -> int:
    """Per-call dropout stream seed drawn from torch's CPU generator (torch.manual_seed controls it)."""
    return int(torch.randint(0, 2 ** 62, (1,)).item())


def _wb(lin):
    if lin is None:
        return None, None
    return lin.weight, lin.bias


class MPN(nn.Module):
    """Bond-message D-MPNN encoder (reference models/mpn.py:11-124)."""

    def __init__(self, bond_fdim: int, atom_fdim: int, MPN_hidden_size: int, MPN_bias: bool = True,
                 MPN_depth: int = 6, MPN_dropout: float = 0.2, return_atom_hiddens: bool = False):
        super().__init__()
        self.return_atom_hiddens = return_atom_hiddens
        self.bond_fdim, self.atom_fdim = bond_fdim, atom_fdim
        self.hidden_size, self.bias, self.depth, self.dropout = MPN_hidden_size, MPN_bias, MPN_depth, MPN_dropout
        self.layers_per_message = 1
        self.cached_zero_vector = nn.Parameter(torch.zeros(self.hidden_size), requires_grad=False)
        self.W_i = nn.Linear(self.bond_fdim, self.hidden_size, bias=self.bias)
        if self.depth > 1:
            self.W_h = nn.Linear(self.hidden_size, self.hidden_size, bias=self.bias)
        self.W_o = nn.Linear(self.atom_fdim + self.hidden_size, self.hidden_size)

    def forward(self, mol_graph, gpu: int, features_batch: List[np.ndarray] = None) -> torch.Tensor:
        g = device_graph_of(mol_graph, gpu)
        if g.f_bonds.shape[1] < self.bond_fdim or g.f_atoms.shape[1] < self.atom_fdim:
            raise RuntimeError("graph feature widths do not match the encoder")
        p = float(self.dropout) if self.training else 0.0
        st = dict(g=g, H=self.hidden_size, depth=self.depth, p=p, seed=_fresh_seed() if p > 0 else 0)
        wh, bh = _wb(getattr(self, "W_h", None))
        h = Fn.MPNFn.apply(st, self.W_i.weight, self.W_i.bias, wh, bh, self.W_o.weight, self.W_o.bias)
        if self.return_atom_hiddens:
            return h
        # molecule readout (reference :110-124) — only the unused query encoder takes this path
        return Fn.SegmentMeanFn.apply(h, g, self.hidden_size)


class MPNDiff(nn.Module):
    """Atom-message MPNN over difference features (reference models/mpn.py:127-240)."""

    def __init__(self, atom_fdim: int, bond_fdim: int, MPNDiff_hidden_size: int, MPNDiff_bias: bool = True,
                 MPNDiff_depth: int = 3, MPNDiff_dropout: float = 0.2):
        super().__init__()
        self.atom_fdim, self.bond_fdim = atom_fdim, bond_fdim
        self.hidden_size, self.bias = MPNDiff_hidden_size, MPNDiff_bias
        self.depth, self.dropout = MPNDiff_depth, MPNDiff_dropout
        self.layers_per_message = 1
        self.cached_zero_vector = nn.Parameter(torch.zeros(self.hidden_size), requires_grad=False)
        self.W_i = nn.Linear(self.atom_fdim, self.hidden_size, bias=self.bias)
        if self.depth > 1:
            self.W_h = nn.Linear(self.hidden_size + self.bond_fdim, self.hidden_size, bias=self.bias)
        if self.depth > 0:
            self.W_o = nn.Linear(self.atom_fdim + self.hidden_size, self.hidden_size)

    def forward(self, atom_features: torch.Tensor, mol_graph, gpu: int,
                features_batch: List[np.ndarray] = None) -> torch.Tensor:
        g = device_graph_of(mol_graph, gpu)
        feat, F = None, 0
        if features_batch is not None:
            feat = torch.as_tensor(np.asarray(features_batch), dtype=torch.float32).reshape(g.M, -1)
            feat = feat.to(g.device).contiguous()
            F = feat.shape[1]
        p = float(self.dropout) if self.training else 0.0
        st = dict(g=g, H=self.hidden_size, depth=self.depth, p=p, seed=_fresh_seed() if p > 0 else 0, feat=feat, F=F)
        wh, bh = _wb(getattr(self, "W_h", None))
        wo, bo = _wb(getattr(self, "W_o", None))
        return Fn.MPNDiffFn.apply(st, atom_features, self.W_i.weight, self.W_i.bias, wh, bh, wo, bo)
