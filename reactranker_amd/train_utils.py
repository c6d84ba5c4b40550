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


class HipAdam(Optimizer):
    """torch.optim.Adam's update (same formula, same defaults, same `state_dict` layout: step / exp_avg / exp_avg_sq per
    parameter) issued as ONE launch of rr_adam_step_f32 over all of a group's tensors.  torch's fused Adam hands its
    multi-tensor kernel 64k-element chunks - about 20 workgroups for this model's 0.8 M parameters, 45 us + 5 us for the
    step counters on MI355X, inside the serial stretch between backward and the next forward; here a workgroup takes 1024
    elements (~8 us).  Parameters without a gradient are skipped like torch skips them.  fp32 CUDA parameters only;
    amsgrad / maximize / capturable are not offered (the reference uses none of them, train/utils.py:100-113)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0 and eps >= 0.0 and lr >= 0.0 and weight_decay >= 0.0):
            raise ValueError("HipAdam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        import ctypes as C
        from . import _lib
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            todo = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise RuntimeError("HipAdam: fp32 contiguous CUDA parameters only")
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)                      # (host tensor, like torch's non-capturable Adam)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                todo.append((p, g, st))
            b1, b2 = group["betas"]
            for i in range(0, len(todo), _lib.RR_MAX_ADAM):
                part = todo[i:i + _lib.RR_MAX_ADAM]
                steps = {int(st["step"]) for _, _, st in part}
                for k in steps:                                          # one launch per distinct step count (normally one)
                    sel = [x for x in part if int(x[2]["step"]) == k]
                    arr = (_lib.AdamTensor * len(sel))()
                    for j, (p, g, st) in enumerate(sel):
                        arr[j].p, arr[j].g = p.data_ptr(), g.data_ptr()
                        arr[j].m, arr[j].v, arr[j].n = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()
                    _lib.check(_lib.lib().rr_adam_step_f32(C.cast(arr, C.c_void_p), len(sel), k, float(group["lr"]), float(b1),
                                                           float(b2), float(group["eps"]), float(group["weight_decay"]),
                                                           _lib.stream()), "rr_adam_step_f32")
        return loss


def build_optimizer(model: torch.nn.Module, freeze: bool = False, fused: bool = None) -> Optimizer:
    """Adam(lr=1e-4, weight_decay=0) over the model's (trainable, if freeze) parameters, as train/utils.py:100-113.
    On GPU parameters the update runs as one launch of the library's Adam kernel (HipAdam: torch.optim.Adam's formula and
    state layout); `fused=True` selects torch's own fused Adam instead, `fused=False` its multi-kernel one (also the
    choice for CPU parameters)."""
    params = [p for p in model.parameters() if (p.requires_grad or not freeze)]
    on_gpu = bool(params) and all(p.is_cuda and p.dtype == torch.float32 for p in params)
    if fused is None and on_gpu:
        return HipAdam([{"params": params, "lr": 0.0001, "weight_decay": 0}])
    return Adam([{"params": params, "lr": 0.0001, "weight_decay": 0}], fused=bool(fused) and on_gpu)


def build_lr_scheduler(optimizer: Optimizer, warmup_epochs, total_epochs: int, train_data_size: int, batch_size: int,
                       init_lr: float, max_lr: float, final_lr: float) -> NoamLR:
    return NoamLR(optimizer=optimizer, warmup_epochs=warmup_epochs, total_epochs=total_epochs,
                  steps_per_epoch=train_data_size // batch_size, init_lr=init_lr, max_lr=max_lr, final_lr=final_lr)
