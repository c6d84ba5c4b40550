"""Optimizer / learning-rate plumbing of the reference trainer with the reference's names and arguments
(reactranker/train/utils.py: NoamLR :7-88, param_count :90-97, build_optimizer :100-113, build_lr_scheduler
:116-141), so `main.py`-style drivers run unchanged against reactranker_amd models.  Pinned to the reference
classes themselves by tests/golden/train_utils.npz (tools/make_golden.py --only train_utils)."""
from __future__ import annotations

from typing import List

import torch
from torch.optim import Adam, Optimizer


def noam_lr(step: int, warmup_steps: int, total_steps: int, init_lr: float, max_lr: float, final_lr: float) -> float:
    """Learning rate after `step` scheduler steps: linear warm-up init_lr -> max_lr over warmup_steps, exponential
    decay max_lr -> final_lr until total_steps, final_lr afterwards (train/utils.py:75-84)."""
    if step <= warmup_steps:
        return init_lr + step * ((max_lr - init_lr) / warmup_steps)
    if step <= total_steps:
        gamma = (final_lr / max_lr) ** (1 / (total_steps - warmup_steps))
        return max_lr * (gamma ** (step - warmup_steps))
    return final_lr


class NoamLR:
    """NoamLR(optimizer, warmup_epochs, total_epochs, steps_per_epoch, init_lr, max_lr, final_lr).

    Same observable behaviour as the reference scheduler: construction already takes one step (the reference
    derives from torch's _LRScheduler, whose constructor calls step()), only `param_groups[0]['lr']` is written
    (train/utils.py:88), `step(current_step=k)` jumps to step k."""

    def __init__(self, optimizer: Optimizer, warmup_epochs, total_epochs: int, steps_per_epoch: int, init_lr: float,
                 max_lr: float, final_lr: float):
        self.optimizer = optimizer
        self.warmup_epochs, self.total_epochs, self.steps_per_epoch = warmup_epochs, total_epochs, steps_per_epoch
        self.init_lr, self.max_lr, self.final_lr = init_lr, max_lr, final_lr
        self.warmup_steps = int(warmup_epochs * steps_per_epoch)
        self.total_steps = total_epochs * steps_per_epoch
        self.current_step = 0
        self.lr = init_lr
        self.step()

    def get_lr(self) -> List[float]:
        return [self.lr]

    def step(self, current_step: int = None) -> None:
        self.current_step = self.current_step + 1 if current_step is None else current_step
        self.lr = noam_lr(self.current_step, self.warmup_steps, self.total_steps, self.init_lr, self.max_lr, self.final_lr)
        self.optimizer.param_groups[0]["lr"] = self.lr

    def state_dict(self) -> dict:
        return {"current_step": self.current_step, "lr": self.lr}

    def load_state_dict(self, state: dict) -> None:
        self.step(current_step=int(state["current_step"]))


def param_count(model: torch.nn.Module) -> int:
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def build_optimizer(model: torch.nn.Module, freeze: bool = False, fused: bool = None) -> Optimizer:
    """Adam(lr=1e-4, weight_decay=0) over the model's (trainable, if freeze) parameters, as train/utils.py:100-113.
    `fused` (default: on for GPU parameters) selects torch's single-kernel Adam - the same update in one launch
    instead of eight, 4 % of a training step on MI355X."""
    params = [p for p in model.parameters() if (p.requires_grad or not freeze)]
    if fused is None:
        fused = bool(params) and all(p.is_cuda for p in params)
    return Adam([{"params": params, "lr": 0.0001, "weight_decay": 0}], fused=fused)


def build_lr_scheduler(optimizer: Optimizer, warmup_epochs, total_epochs: int, train_data_size: int, batch_size: int,
                       init_lr: float, max_lr: float, final_lr: float) -> NoamLR:
    return NoamLR(optimizer=optimizer, warmup_epochs=warmup_epochs, total_epochs=total_epochs,
                  steps_per_epoch=train_data_size // batch_size, init_lr=init_lr, max_lr=max_lr, final_lr=final_lr)
