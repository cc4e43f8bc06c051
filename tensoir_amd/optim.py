"""optimizer.step() of the training loop on one HIP launch.

The reference's loop (train_tensoIR.py:197, :315-317) builds `torch.optim.Adam(grad_vars, betas=(0.9, 0.99))` over ~35
parameter tensors in per-tensor groups and calls `optimizer.step()` after `total_loss.backward()`.  torch's multi-tensor
Adam spends ~90 launches per step on that list; `Adam` below is the same optimizer (same constructor, same state_dict
entries `step` / `exp_avg` / `exp_avg_sq`, same update arithmetic for its default mode) whose `step()` is
`tir_adam_step`: one pass over parameter, gradient and both moments.  `tensoir_amd.run` binds `torch.optim.Adam` to it.

Modes the kernel does not implement (amsgrad, weight decay, maximize, capturable/differentiable, sparse gradients,
non-fp32 or non-CUDA parameters) raise -- nothing falls back silently.
"""
from __future__ import annotations

import functools

import torch

from . import ops
from ._lib import TensoirHipError

_TorchAdam = torch.optim.Adam


@functools.lru_cache(maxsize=512)
def _dense_key_of(shape, stride):
    dims = sorted((st, sz) for sz, st in zip(shape, stride) if sz > 1)
    expect = 1
    for st, sz in dims:
        if st != expect:
            return None
        expect *= sz
    return tuple(st for sz, st in zip(shape, stride) if sz > 1)


def _dense_key(t):
    """Strides over the dims of extent > 1 if `t` is non-overlapping and dense, else None."""
    return _dense_key_of(tuple(t.shape), t.stride())


class Adam(_TorchAdam):
    """torch.optim.Adam with a single-launch HIP `step()` (see the module docstring)."""

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # pass 1 validates EVERY group and parameter and mutates nothing: an unsupported option or tensor on a later
        # parameter must not leave earlier ones with an advanced step count and no update
        todo = []
        betas = eps = None
        for group in self.param_groups:
            for opt in ("amsgrad", "maximize", "capturable", "differentiable"):
                if group.get(opt):
                    raise NotImplementedError(f"tensoir_amd.optim.Adam: {opt}=True is not implemented")
            if group.get("weight_decay", 0) != 0:
                raise NotImplementedError("tensoir_amd.optim.Adam: weight_decay != 0 is not implemented")
            b, e = tuple(float(x) for x in group["betas"]), float(group["eps"])
            if betas is None:
                betas, eps = b, e
            elif (b, e) != (betas, eps):
                raise NotImplementedError("tensoir_amd.optim.Adam: betas / eps must be the same in every parameter group")
        for group in self.param_groups:
            lr = float(group["lr"])
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                if g.is_sparse or p.dtype != torch.float32 or g.dtype != torch.float32:
                    raise NotImplementedError("tensoir_amd.optim.Adam: dense fp32 parameters and gradients only")
                if not p.is_cuda:
                    raise TensoirHipError("tensoir_amd.optim.Adam.step needs the parameters on an MI355X (no CPU path)")
                key = _dense_key_of(tuple(p.shape), p.stride())
                if key is None:
                    raise NotImplementedError("tensoir_amd.optim.Adam: parameters must be non-overlapping and dense")
                todo.append((p, g, key, lr))
        # pass 2: state creation / layout fixes / step counts, then the one launch
        entries = []
        for p, g, key, lr in todo:
            ps = p.stride()
            if g.stride() != ps and _dense_key(g) != key:
                g = torch.empty_like(p).copy_(g)          # autograd's layout contract makes this the rare case
            state = self.state[p]
            if len(state) == 0:
                state["step"] = torch.tensor(0.0, dtype=torch.float32)
                state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            m, v = state["exp_avg"], state["exp_avg_sq"]
            if (m.stride() != ps and _dense_key(m) != key) or (v.stride() != ps and _dense_key(v) != key):      # e.g. a state_dict loaded into another layout
                m = state["exp_avg"] = torch.empty_like(p).copy_(m)
                v = state["exp_avg_sq"] = torch.empty_like(p).copy_(v)
            state["step"] += 1
            t = float(state["step"])
            entries.append((p, g, m, v, lr, 1.0 - betas[0] ** t, 1.0 - betas[1] ** t))
        if entries:
            ops.adam_step(entries, betas[0], betas[1], eps)
            # the kernel wrote parameters and moments through raw pointers: tell autograd (saved-tensor checks) and every
            # cache keyed by Tensor._version (the model's packed decoder images, light means, descriptors) that they changed
            torch.autograd.graph.increment_version([t for e in entries for t in (e[0], e[2], e[3])])
        return loss


def _supported(opt) -> bool:
    """True when every parameter group of `opt` is something the single-launch step implements (see Adam.step)."""
    betas = eps = None
    for group in opt.param_groups:
        if any(group.get(o) for o in ("amsgrad", "maximize", "capturable", "differentiable")) or group.get("weight_decay", 0) != 0:
            return False
        b, e = tuple(float(x) for x in group["betas"]), float(group["eps"])
        if betas is None:
            betas, eps = b, e
        elif (b, e) != (betas, eps):
            return False
        for p in group["params"]:
            if not p.is_cuda or p.dtype != torch.float32 or _dense_key(p) is None:
                return False
            if p.grad is not None and (p.grad.is_sparse or p.grad.dtype != torch.float32):
                return False
    return True


class LauncherAdam(Adam):
    """What `python -m tensoir_amd.run` binds to the name torch.optim.Adam: the training script's optimizer (fp32 CUDA
    parameters, default options -- train_tensoIR.py:197) takes the single-launch step; any OTHER Adam the process creates
    (weight decay, amsgrad, CPU parameters: a metric network's, a user's own code) behaves exactly as torch.optim.Adam does.
    The decision is made per step from the optimizer's own groups, so the rebinding never changes what foreign code gets."""

    @torch.no_grad()
    def step(self, closure=None):
        if _supported(self):
            return Adam.step(self, closure)
        return _TorchAdam.step(self, closure)
