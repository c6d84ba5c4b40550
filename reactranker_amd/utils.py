"""Mirror of the hot-path pieces of reference `reactranker/utils.py`:
index_select_ND (:176-193) and the checkpoint format (:152-173)."""
import os

import torch

from . import functions as Fn


def index_select_ND(source: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """source[index] for a 2-D index -> [n, K, H] (reference utils.py:176-193), differentiable in `source`.

    Kept for API parity; the model itself never materialises this tensor — it calls
    functions.gather_sum, which fuses the following .sum(dim=1).
    """
    return Fn.IndexSelectNDFn.apply(source, index)


def index_select_sum(source: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """index_select_ND(source, index).sum(dim=1) as one differentiable fused op."""
    return Fn.GatherSumFn.apply(source, index)


def save_checkpoint(path: str, model, means=None, stds=None) -> None:
    """Same dict layout as reference utils.py:152-173."""
    state = {
        'state_dict': model.state_dict(),
        'data_scaler': {'means': means, 'stds': stds} if means is not None and stds is not None else None,
    }
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    torch.save(state, path)


def load_checkpoint(path: str, model, map_location=None):
    state = torch.load(path, map_location=map_location, weights_only=False)
    model.load_state_dict(state['state_dict'])
    return state.get('data_scaler')


def param_count(model) -> int:
    return sum(p.numel() for p in model.parameters() if p.requires_grad)
