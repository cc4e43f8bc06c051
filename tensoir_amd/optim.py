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
import os

import torch

from . import ops
from ._lib import TensoirHipError

_TorchAdam = torch.optim.Adam
FAST_STEP = os.environ.get("TENSOIR_FAST_ADAM", "1") != "0"        # 0: every step takes the fully checked path


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


class _Rec:
    """What `Adam.step` knows about one parameter after a first (fully checked) step."""
    __slots__ = ("p", "ptr", "stride", "state", "step_t", "step_np", "m", "v", "m_ptr", "v_ptr")


class _Plan:
    """The optimizer's parameter list as of its last fully checked step: static pointer tables of the one-launch kernel and the
    identities that must still hold for them to be valid (see Adam._fast_step)."""
    __slots__ = ("state_obj", "recs", "betas", "eps", "tables", "touched")


class Adam(_TorchAdam):
    """torch.optim.Adam with a single-launch HIP `step()` (see the module docstring)."""

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not self._fast_step():
            self._checked_step()
        return loss

    def _fast_step(self) -> bool:
        """The steady state of a training loop: the same parameters, moments and options as in the last fully checked step, fresh
        gradients.  The host side of a step is then ~35 identity checks and the gradient pointers (the loop is host-bound: the
        checked path costs 0.35 ms per step on the reference's ~35 per-tensor groups).  False = something differs (a parameter
        without gradient, another layout, a reloaded state, an option) and NOTHING has been changed: the caller takes the checked
        path, which also rebuilds the plan."""
        plan = self.__dict__.get("_tir_plan")
        if plan is None or plan.state_obj is not self.state or not FAST_STEP:
            return False
        recs, betas, eps = plan.recs, plan.betas, plan.eps
        n_rec, i = len(recs), 0
        f32, strided = torch.float32, torch.strided
        grads, lrs = [], []
        for group in self.param_groups:
            if (group.get("amsgrad") or group.get("maximize") or group.get("capturable") or group.get("differentiable")
                    or group.get("weight_decay", 0) != 0 or group["eps"] != eps):
                return False
            b = group["betas"]
            if b is not betas and tuple(b) != betas:
                return False
            lr = float(group["lr"])
            for p in group["params"]:
                if i >= n_rec:
                    return False
                rec = recs[i]
                i += 1
                g = p.grad
                if (rec.p is not p or g is None or g.dtype is not f32 or g.layout is not strided or not g.is_cuda
                        or g.stride() != rec.stride or p.stride() != rec.stride or p.data_ptr() != rec.ptr):
                    return False
                st = rec.state
                m, v = st.get("exp_avg"), st.get("exp_avg_sq")
                if st.get("step") is not rec.step_t or m is not rec.m or v is not rec.v:
                    return False
                # the same tensor objects may have been given other storage (`m.data = ...`, `set_`): the kernel's table holds
                # raw pointers, so storage and layout are re-checked too
                if m.data_ptr() != rec.m_ptr or v.data_ptr() != rec.v_ptr or m.stride() != rec.stride or v.stride() != rec.stride:
                    return False
                grads.append(g)
                lrs.append(lr)
        if i != n_rec:
            return False
        # ---- every check has passed: step counts, the three per-step columns of the kernel's table, the launch ----
        PA_g, F_lr, F_b1, F_b2 = plan.tables[4:]
        b1, b2 = betas
        corr = {}
        for j, rec in enumerate(recs):
            t = float(rec.step_np) + 1.0
            c = corr.get(t)
            if c is None:
                c = corr[t] = (1.0 - b1 ** t, 1.0 - b2 ** t)
            PA_g[j] = grads[j].data_ptr()
            F_lr[j] = lrs[j]
            F_b1[j], F_b2[j] = c
        ops.adam_step_tables(n_rec, plan.tables, b1, b2, eps)      # raises -> no step count has moved, nothing was updated
        for rec in recs:
            rec.step_np[()] = float(rec.step_np) + 1.0    # state["step"] += 1, through the tensor's own memory
        torch.autograd.graph.increment_version(plan.touched)
        return True

    def _checked_step(self):
        # pass 1 validates EVERY group and parameter and mutates nothing: an unsupported option or tensor on a later
        # parameter must not leave earlier ones with an advanced step count and no update
        self.__dict__["_tir_plan"] = None
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
        n_params = 0
        for group in self.param_groups:
            lr = float(group["lr"])
            for p in group["params"]:
                n_params += 1
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
        plain = True                    # every gradient and moment already in its parameter's own layout (what the plan assumes)
        for p, g, key, lr in todo:
            ps = p.stride()
            if g.stride() != ps:
                plain = False
                if _dense_key(g) != key:
                    g = torch.empty_like(p).copy_(g)          # autograd's layout contract makes this the rare case
            state = self.state[p]
            if len(state) == 0:
                state["step"] = torch.tensor(0.0, dtype=torch.float32)
                state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            m, v = state["exp_avg"], state["exp_avg_sq"]
            if m.stride() != ps or v.stride() != ps:
                plain = False
                if _dense_key(m) != key or _dense_key(v) != key:      # e.g. a state_dict loaded into another layout
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
        if plain and entries and len(entries) == n_params:
            self._make_plan(entries, betas, eps)

    def _make_plan(self, entries, betas, eps):
        """After a fully checked step over ALL parameters in their plain layout: remember them for _fast_step."""
        recs = []
        for p, _g, m, v, *_ in entries:
            st = self.state[p]
            step_t = st["step"]
            if not torch.is_tensor(step_t) or step_t.is_cuda or step_t.dtype != torch.float32 or step_t.dim() != 0 or step_t.requires_grad:
                return
            rec = _Rec()
            rec.p, rec.ptr, rec.stride, rec.state = p, p.data_ptr(), p.stride(), st
            rec.step_t, rec.step_np, rec.m, rec.v = step_t, step_t.numpy(), m, v
            rec.m_ptr, rec.v_ptr = m.data_ptr(), v.data_ptr()
            recs.append(rec)
        plan = _Plan()
        plan.state_obj, plan.recs, plan.betas, plan.eps = self.state, recs, betas, eps
        plan.tables = ops.adam_tables([(r.p, r.m, r.v) for r in recs])
        plan.touched = [t for r in recs for t in (r.p, r.m, r.v)]
        self.__dict__["_tir_plan"] = plan


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
        # the launcher's data-parallel mode (dist.LAUNCHER_DP: the unmodified script under torchrun): average the gradients over
        # the ranks first -- bucketed all-reduce in the parameters' own memory order; every rank issues the same collectives
        from . import dist as tdist
        if tdist.launcher_dp() is not None:
            if closure is not None:
                with torch.enable_grad():
                    closure()
                closure = None
            tdist.allreduce_gradients([p for g in self.param_groups for p in g["params"]])
        # the steady state first: a plan only exists after a step that _supported() accepted, and Adam._fast_step re-checks
        # every condition of _supported() that can change between two steps
        if closure is None and self.__dict__.get("_tir_plan") is not None and self._fast_step():
            return None
        if _supported(self):
            return Adam.step(self, closure)
        self.__dict__["_tir_plan"] = None
        return _TorchAdam.step(self, closure)
