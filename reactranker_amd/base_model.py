"""FFN / ReactionModel / build_model — mirror of reference `reactranker/models/base_model.py`.

`build_model(...)` keeps the reference signature (:235-245) and returns a module whose
state_dict keys and shapes equal the reference's, so reference
checkpoints load unchanged.  `model(r_inputs, p_inputs, gpu=, add_features=)` follows
ReactionModel.forward (:150-171); the whole forward/backward runs as one autograd node
(functions.ReactionModelFn) on the HIP kernels.
"""
from typing import List

import numpy as np
import torch
import torch.nn as nn

from . import functions as Fn
from .featurization import ATOM_FDIM, BOND_FDIM, device_graph_of
from .mpn import MPN, MPNDiff, _fresh_seed, _wb


class FFN(nn.Module):
    """Reference models/base_model.py:10-108 (same Sequential layout -> keys ffn.ffn.{1,4,7})."""

    def __init__(self, reacvec_fdim: int, ffn_hidden_size: int, ffn_dropout: float = 0.2, ffn_num_layers: int = 3,
                 task_num: int = 2, ffn_bias: bool = True, task_type: str = 'gaussian'):
        super().__init__()
        self.hidden_size, self.ffn_hidden_size = reacvec_fdim, ffn_hidden_size
        self.dropout, self.ffn_num_layers = ffn_dropout, ffn_num_layers
        self.activation = nn.ReLU()
        self.task_type, self.bias, self.output = task_type, ffn_bias, None
        self.task_num = task_num
        dropout = nn.Dropout(self.dropout)
        activation = self.activation
        if self.ffn_num_layers == 1:
            ffn = [dropout, nn.Linear(self.hidden_size, task_num, bias=self.bias)]
        else:
            ffn = [dropout, nn.Linear(self.hidden_size, self.ffn_hidden_size, bias=self.bias)]
            for _ in range(self.ffn_num_layers - 2):
                ffn.extend([activation, dropout, nn.Linear(self.ffn_hidden_size, self.ffn_hidden_size, bias=self.bias)])
            ffn.extend([activation, dropout, nn.Linear(self.ffn_hidden_size, task_num, bias=self.bias)])
        self.ffn = nn.Sequential(*ffn)

    def linears(self):
        return [m for m in self.ffn if isinstance(m, nn.Linear)]

    def head(self) -> int:
        h = Fn.HEADS.get(self.task_type, 0)
        n = self.task_num
        if (h in (3, 4, 5) and n % 2) or (h == 6 and n % 4):
            raise RuntimeError(f"task_type {self.task_type!r} needs task_num divisible by its parameter count")
        return h

    def flat_params(self):
        out = []
        for lin in self.linears():
            out += [lin.weight, lin.bias]
        return out

    def forward(self, x):
        p = float(self.dropout) if self.training else 0.0
        st = dict(p=p, seed=_fresh_seed() if p > 0 else 0, head=self.head(), squeeze=(self.task_num == 1))
        self.output = Fn.FFNFn.apply(st, x, *self.flat_params())
        return self.output


class ReactionModel(nn.Module):
    """Reference models/base_model.py:111-171."""

    def __init__(self, mpnn_hidden_size: int = 300, mpnn_bias: bool = True, mpnn_depth: int = 3, mpnn_dropout=0.2,
                 mpnn_diff_hidden_size: int = 300, mpnn_diff_bias: bool = True, mpnn_diff_depth: int = 3,
                 mpnn_diff_dropout=0.2, ffn_hidden_size: int = 300, ffn_bias: bool = True, ffn_dropout=0.2,
                 ffn_depth: int = 3, task_num: int = 2, task_type: str = 'no_softplus',
                 addtion_react_featrues: int = 0):
        super().__init__()
        self.encoder = MPN(bond_fdim=ATOM_FDIM + BOND_FDIM, atom_fdim=ATOM_FDIM, MPN_hidden_size=mpnn_hidden_size,
                           MPN_bias=mpnn_bias, MPN_depth=mpnn_depth, MPN_dropout=mpnn_dropout,
                           return_atom_hiddens=True)
        self.diff_encoder = MPNDiff(atom_fdim=mpnn_hidden_size, bond_fdim=ATOM_FDIM + BOND_FDIM,
                                    MPNDiff_hidden_size=mpnn_diff_hidden_size, MPNDiff_bias=mpnn_diff_bias,
                                    MPNDiff_depth=mpnn_diff_depth, MPNDiff_dropout=mpnn_diff_dropout)
        self.ffn = FFN(reacvec_fdim=mpnn_diff_hidden_size + addtion_react_featrues, ffn_hidden_size=ffn_hidden_size,
                       ffn_dropout=ffn_dropout, ffn_num_layers=ffn_depth, task_num=task_num, ffn_bias=ffn_bias,
                       task_type=task_type)
        if mpnn_hidden_size != mpnn_diff_hidden_size:
            raise NotImplementedError("build_model always ties the hidden sizes (reference base_model.py:266-280)")
        self.dropout_seed = None     # set to an int to pin the dropout streams (tests)
        # Reactant de-duplication: every candidate of a query repeats the same reactant
        # graph.  "auto" (default): with dropout inactive (eval or p == 0) encoder(r) runs once per distinct
        # reactant; in train mode with dropout only its deterministic prefix (everything before the first
        # dropout) is shared and each copy keeps its own mask stream.  Both are exact.  False: never share.
        self.dedup_reactants = "auto"

    def flat_params(self):
        e, d = self.encoder, self.diff_encoder
        out = [e.W_i.weight, e.W_i.bias, *_wb(getattr(e, "W_h", None)), e.W_o.weight, e.W_o.bias,
               d.W_i.weight, d.W_i.bias, *_wb(getattr(d, "W_h", None)), *_wb(getattr(d, "W_o", None))]
        return out + self.ffn.flat_params()

    def forward(self, r_inputs, p_inputs, gpu: int = None, add_features: List[np.ndarray] = None):
        rg = device_graph_of(r_inputs, gpu)
        pg = device_graph_of(p_inputs, gpu)
        dedup = None
        p_active = float(self.encoder.dropout) if self.training else 0.0
        want = self.dedup_reactants in ("auto", True) and p_active == 0.0
        if want and hasattr(r_inputs, "unique"):
            ub, amap, amap_t = r_inputs.unique()
            if ub.n_mols < r_inputs.n_mols:
                cache = getattr(r_inputs, "_rr_dedup_dev", None)
                if cache is None or cache[0] != str(pg.device):
                    cache = (str(pg.device), torch.from_numpy(amap).to(pg.device), torch.from_numpy(amap_t).to(pg.device))
                    r_inputs._rr_dedup_dev = cache
                dedup = (ub.device_graph(pg.device), cache[1], cache[2])
        prefix = None
        if (dedup is None and self.dedup_reactants in ("auto", True) and p_active > 0.0 and self.encoder.depth >= 2
                and self.encoder.hidden_size % 4 == 0            # the shared-prefix backward needs the fused dZ side output
                and hasattr(r_inputs, "unique_bonds")):
            # train mode with dropout: only the deterministic prefix of the reactant encoder (up to its first
            # dropout) is shared across the copies — exact, the per-copy mask stream is unchanged
            ub, _, _ = r_inputs.unique()
            if ub.n_mols < r_inputs.n_mols:
                cache = getattr(r_inputs, "_rr_prefix_dev", None)
                if cache is None or cache[0] != str(pg.device):
                    bmap, bmap_t = r_inputs.unique_bonds()
                    cache = (str(pg.device), torch.from_numpy(bmap).to(pg.device), torch.from_numpy(bmap_t).to(pg.device))
                    r_inputs._rr_prefix_dev = cache
                prefix = (ub.device_graph(pg.device), cache[1], cache[2])
        if rg.nA != pg.nA:
            raise RuntimeError("reactant and product batches must hold the same atoms in the same order "
                               "(diff = p_h - r_h, reference models/base_model.py:168)")
        feat, F = None, 0
        if add_features is not None:
            if torch.is_tensor(add_features):
                feat = add_features.to(device=pg.device, dtype=torch.float32).reshape(pg.M, -1).contiguous()
            else:
                feat = torch.as_tensor(np.asarray(add_features), dtype=torch.float32).reshape(pg.M, -1)
                feat = feat.to(pg.device).contiguous()
            F = feat.shape[1]
        if self.diff_encoder.hidden_size + F != self.ffn.hidden_size:
            raise RuntimeError(f"add_features width {F} does not match the FFN input "
                               f"({self.ffn.hidden_size - self.diff_encoder.hidden_size})")
        drops = {float(self.encoder.dropout), float(self.diff_encoder.dropout), float(self.ffn.dropout)}
        if len(drops) != 1:
            raise NotImplementedError("build_model uses one dropout rate everywhere (reference base_model.py:266-280)")
        p = drops.pop() if self.training else 0.0
        seed = 0
        if p > 0:
            seed = _fresh_seed() if self.dropout_seed is None else int(self.dropout_seed)
        st = dict(r=rg if dedup is None else dedup[0], dedup=dedup, prefix=prefix, p_graph=pg, H=self.encoder.hidden_size, depth=self.encoder.depth,
                  diff_depth=self.diff_encoder.depth, p=p, seed=seed, feat=feat, F=F, head=self.ffn.head(),
                  squeeze=(self.ffn.task_num == 1))
        out = Fn.ReactionModelFn.apply(st, *self.flat_params())
        self.ffn.output = out
        return out


def build_model(hidden_size: int = 300, mpnn_depth: int = 3, mpnn_diff_depth: int = 3, ffn_depth: int = 3,
                use_bias: bool = True, dropout=0.2, task_num: int = 2, ffn_last_layer: str = 'no_softplus',
                task_type=None, bimolecule=False, add_features_dim=0):
    """Reference models/base_model.py:235-297 (same head-string logic :252-264)."""
    if task_type is None:
        if task_num == 2:
            task_type = 'gaussian_' + ffn_last_layer
        elif task_num == 4:
            task_type = 'evidential_' + ffn_last_layer
        else:
            task_type = ffn_last_layer
    elif task_type == 'evidential_ranking':
        task_type = task_type
    else:
        task_type = task_type + '_' + ffn_last_layer
    return ReactionModel(mpnn_hidden_size=hidden_size, mpnn_bias=use_bias, mpnn_depth=mpnn_depth,
                         mpnn_dropout=dropout, mpnn_diff_hidden_size=hidden_size, mpnn_diff_bias=use_bias,
                         mpnn_diff_depth=mpnn_diff_depth, mpnn_diff_dropout=dropout, ffn_hidden_size=hidden_size,
                         ffn_bias=use_bias, ffn_dropout=dropout, ffn_depth=ffn_depth, task_num=task_num,
                         task_type=task_type, addtion_react_featrues=0 if bimolecule else add_features_dim)
