"""Optimizer of the reference's training loop on the HIP path.

train.py:52-57 builds ``optim.Adam(pg0, lr, betas=(0.9, 0.999))`` and adds two more parameter groups (weights with weight decay,
biases); train.py:138 calls ``optimizer.step()`` once per pair.  ``Adam`` below is that optimizer with the same constructor, the same
``param_groups`` / ``add_param_group`` / ``state_dict`` (state per parameter: ``step``, ``exp_avg``, ``exp_avg_sq`` -- a checkpoint
written by ``torch.optim.Adam`` loads and vice versa) and the same arithmetic, but ONE fused multi-tensor launch sequence per step
(``gims_adam_step``, csrc/optim.hip) instead of torch's per-operation list kernels: the 282 tensors of a GMatcher take 4 launches.

    from gims_amd.optim import Adam            # instead of optim.Adam in train.py:53
"""
from __future__ import annotations

import numpy as np
import torch

from . import hip

__all__ = ["Adam"]


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam's interface and update rule (L2 weight decay added to the gradient, bias-corrected moments); ``amsgrad``,
    ``maximize``, ``capturable``, ``differentiable`` and sparse gradients are not built and raise.  Parameters must be float32 tensors
    on the GPU: there is no CPU path."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, *, maximize=False, foreach=None,
                 capturable=False, differentiable=False, fused=None):
        if amsgrad or maximize or capturable or differentiable:
            raise NotImplementedError("gims_amd.optim.Adam: amsgrad / maximize / capturable / differentiable are not built")
        if isinstance(lr, torch.Tensor):
            raise NotImplementedError("gims_amd.optim.Adam: lr must be a Python number")
        if not 0.0 <= lr:
            raise ValueError(f"Invalid learning rate: {lr}")
        if not 0.0 <= eps:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 0: {betas[0]}")
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 1: {betas[1]}")
        if not 0.0 <= weight_decay:
            raise ValueError(f"Invalid weight_decay value: {weight_decay}")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False, foreach=None,
                                      capturable=False, differentiable=False, fused=None))

        self._plan = None

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._plan = None

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._plan = None

    def _make_plan(self, active):
        """Everything about a step that does not change while the same parameters receive gradients: the pointer table (only its
        `grad` column is refreshed per step), the step tensors, and the (param group, step count) classes that share one set of
        bias corrections."""
        classes, cls_of, rows, counts = [], {}, [], []
        for gi, p in active:
            if not p.is_cuda or p.dtype != torch.float32:
                raise RuntimeError("gims_amd.optim.Adam needs float32 parameters on the GPU (no CPU path)")
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)               # torch's layout: a CPU scalar tensor per parameter
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if not torch.is_tensor(st["step"]):
                st["step"] = torch.tensor(float(st["step"]), dtype=torch.float32)
            if st["step"].is_cuda:
                raise NotImplementedError("gims_amd.optim.Adam: capturable state (step on the GPU) is not built")
            m, v = st["exp_avg"], st["exp_avg_sq"]
            if not (p.is_contiguous() and m.is_contiguous() and v.is_contiguous()) or m.dtype != torch.float32 or v.dtype != torch.float32 or not m.is_cuda:
                raise RuntimeError("gims_amd.optim.Adam needs contiguous float32 parameters and moments on the GPU")
            key = (gi, int(st["step"].item()))
            if key not in cls_of:
                cls_of[key] = len(classes)
                classes.append([gi, key[1]])
            counts.append(float(key[1]))
            rows.append((p.data_ptr(), 0, m.data_ptr(), v.data_ptr(), p.numel(), cls_of[key], 0))
        table = np.array(rows, dtype=hip.ADAM_TENSOR_DTYPE)
        # the per-parameter step counts (CPU scalars in torch's layout) become views into ONE vector, so that a step increments them
        # with one operation (282 separate scalars: 0.4 ms per step); distinct elements -- an in-place update through any one of
        # them, e.g. by torch.optim.Adam after loading this optimizer's state_dict, touches only its own.  A parameter that drops
        # out of the active set keeps its view into the vector of the plan it was last part of.
        steps = torch.tensor(counts, dtype=torch.float32)
        for i, (_, p) in enumerate(active):
            self.state[p]["step"] = steps[i]
        return dict(ids=[id(p) for _, p in active], ptrs=table["param"].tolist(), table=table, classes=classes, steps=steps)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        active, grads = [], []
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                g = p.grad
                if g is not None:
                    active.append((gi, p))
                    grads.append(g)
        if not active:
            return loss
        plan = self._plan
        # the plan holds raw pointers: it is rebuilt when the set of parameters with gradients or a parameter's storage changes
        # (load_state_dict / add_param_group drop it; replacing a moment tensor in .state by hand is not detected)
        if plan is None or plan["ids"] != [id(p) for _, p in active] or plan["ptrs"] != [p.data_ptr() for _, p in active]:
            plan = self._plan = self._make_plan(active)
        keep = []
        for i, g in enumerate(grads):
            if g.is_sparse:
                raise RuntimeError("Adam does not support sparse gradients")
            if not g.is_cuda or g.dtype != torch.float32:
                raise RuntimeError("gims_amd.optim.Adam needs float32 gradients on the GPU (no CPU path)")
            if not g.is_contiguous():
                grads[i] = g.contiguous()
                keep.append(grads[i])
        table = plan["table"]
        table["grad"] = [g.data_ptr() for g in grads]
        plan["steps"] += 1
        hyper = []
        for c in plan["classes"]:
            c[1] += 1
            group = self.param_groups[c[0]]
            if group.get("amsgrad") or group.get("maximize"):
                raise NotImplementedError("gims_amd.optim.Adam: amsgrad / maximize are not built")
            hyper.append(dict(lr=group["lr"], beta1=group["betas"][0], beta2=group["betas"][1], eps=group["eps"], weight_decay=group["weight_decay"], step=c[1]))
        if len(hyper) <= 8:
            hip.adam_step(table, hyper)
        else:
            for lo in range(0, len(hyper), 8):
                sel = table[(table["group"] >= lo) & (table["group"] < lo + 8)].copy()
                sel["group"] -= lo
                hip.adam_step(np.ascontiguousarray(sel), hyper[lo:lo + 8])
        return loss
