"""Optimiser side of the drop-in surface: ``Adam`` (what ``load()`` returns, same
interface and ``state_dict`` format as ``torch.optim.Adam`` —
/root/reference/model/bilinear.py:60, train_bilinear.py:75,83,101) and
``clip_grad_norm_`` (/root/reference/train_bilinear.py:81), both executed by the
fused kernels over the flat gradient / moment arenas.
"""
from __future__ import annotations

import torch


def _engine_of(module):
    eng = getattr(module, "engine", None)
    if eng is None:
        raise RuntimeError("bilinear_amd.optim needs a bilinear_amd BilinearUnit")
    return eng


class Adam(torch.optim.Optimizer):
    """Adam with torch.optim.Adam's defaults (betas (0.9, 0.999), eps 1e-8, no weight
    decay, no amsgrad).  ``param_groups[i]['lr']`` is assignable (the lr-decay hook of
    train_bilinear.py:66-70 does that); ``state_dict()`` / ``load_state_dict()`` use
    torch.optim.Adam's layout (per-parameter ``step``, ``exp_avg``, ``exp_avg_sq``),
    so reference checkpoints round-trip."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, module=None):
        if module is None:
            raise RuntimeError("bilinear_amd.optim.Adam(params, module=<BilinearUnit>) is required: "
                               "the fused kernel updates the unit's flat parameter arena")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False,
                        maximize=False, foreach=None, capturable=False, differentiable=False,
                        fused=None, decoupled_weight_decay=False)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise RuntimeError("a single parameter group is supported")
        object.__setattr__(self, "_module", module)
        self._exp_avg = None
        self._exp_avg_sq = None
        self._t = 0
        self._stats = None
        self._step_tensor = None
        self._state_for = None          # the engine's parameter list the state views were built for

    # ---------------------------------------------------------------- moments --
    def _ensure_moments(self, engine):
        """Flat exp_avg / exp_avg_sq arenas; ``self.state[p]`` holds views of them."""
        dev = engine.device
        fresh = self._exp_avg is None or self._exp_avg.device != dev \
            or self._exp_avg.numel() != engine.layout.total
        if fresh:
            m = torch.zeros(engine.layout.total, dtype=torch.float32, device=dev)
            v = torch.zeros_like(m)
            for _, p, off, shape in engine._named_params():
                st = self.state.get(p)
                if st and "exp_avg" in st:           # restored from a checkpoint
                    m[off:off + p.numel()].view(shape).copy_(st["exp_avg"])
                    v[off:off + p.numel()].view(shape).copy_(st["exp_avg_sq"])
                    self._t = max(self._t, int(float(st["step"])))
            self._exp_avg, self._exp_avg_sq = m, v
            self._stats = torch.zeros(2, dtype=torch.float32, device=dev)
        named = engine._named_params()
        if not fresh and self._state_for is named and len(self.state) == len(named):
            return                       # (same Parameter objects as when the state views were made)
        if fresh or any(p not in self.state for _, p, _, _ in named):
            # one shared host scalar: torch.optim.Adam keeps a per-parameter `step`
            # tensor; they are always equal, so every entry aliases this one
            self._step_tensor = torch.tensor(float(self._t))
            for _, p, off, shape in engine._named_params():
                self.state[p] = {
                    "step": self._step_tensor,
                    "exp_avg": self._exp_avg[off:off + p.numel()].view(shape),
                    "exp_avg_sq": self._exp_avg_sq[off:off + p.numel()].view(shape),
                }
        self._state_for = named

    def _sync_step_state(self, engine):
        self._step_tensor.fill_(float(self._t))

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._exp_avg = None        # re-pack from self.state at the next step
        self._exp_avg_sq = None
        self._t = 0
        for st in self.state.values():
            if "step" in st:
                self._t = max(self._t, int(float(st["step"])))

    # ------------------------------------------------------------------ steps --
    def _gather_grads(self, engine):
        """Make sure every parameter's gradient sits in the arena slot."""
        if engine.grads_in_arena():
            return 0
        missing = 0
        for (_, p, off, shape), ptr in zip(engine._named_params(), engine.grad_ptrs()):
            g = p.grad
            if g is None:
                missing += 1
            elif g.data_ptr() != ptr:
                engine.grad_view(off, shape).copy_(g)
        return missing

    def zero_grad(self, set_to_none=True):
        """optimizer.zero_grad() of train_bilinear.py:75.  torch's version walks the groups through a foreach grouping
        under a dynamo-disable wrapper (20 us of host time; the five-call step at batch 64 is host-bound): here the
        22 ``.grad`` fields are dropped (``set_to_none=True``, torch's default) or the gradient arena is zeroed in
        one launch."""
        engine = getattr(self._module, "_engine", None)
        if engine is None or engine.grads is None or torch.compiler.is_compiling():
            return super().zero_grad(set_to_none)
        if set_to_none:
            for _, p, _, _ in engine._named_params(validate=False):
                p.grad = None
        elif engine.grads_in_arena():
            engine.grads.zero_()
        else:
            super().zero_grad(False)

    def step(self, closure=None):
        """optimizer.step() of train_bilinear.py:83.  torch wraps every Optimizer.step in a profiler record + hook
        loop (Optimizer.profile_hook_step: 25 us of host time per call); that wrapper is taken only when a step hook
        is registered — with none, which is the reference's case, the step runs bare."""
        from torch.optim import optimizer as _O
        if (self._optimizer_step_pre_hooks or self._optimizer_step_post_hooks or _O._global_optimizer_pre_hooks
                or _O._global_optimizer_post_hooks):
            return Adam._step_with_hooks(self, closure)
        return self._step_impl(closure)
    step.hooked = True          # (Optimizer._patch_step_function: do not wrap this one)

    @torch.no_grad()
    def _step_impl(self, closure=None):
        if closure is not None:
            raise RuntimeError("closures are not supported")
        engine = _engine_of(self._module)
        if engine.params is None:
            return None
        missing = self._gather_grads(engine)
        if missing == len(engine.layout.entries):
            return None                                  # nothing to do, like torch
        if missing:
            raise RuntimeError("bilinear_amd.optim.Adam needs the gradient of every parameter "
                               "(%d are None)" % missing)
        self._ensure_moments(engine)
        g = self.param_groups[0]
        self._t += 1
        engine.clip_adam(self._exp_avg, self._exp_avg_sq, float(g["lr"]), g["betas"], g["eps"],
                         None, self._t, self._stats)
        self._sync_step_state(engine)
        return None

    @torch.no_grad()
    def fused_train_step(self, module, x, target, max_norm=1.0, loss_out=None):
        """zero_grad + forward + MSELoss + backward + clip_grad_norm_ + step
        (train_bilinear.py:75-83) as one native enqueue; see BilinearUnit.train_step."""
        engine = _engine_of(module)
        if engine.params is None or engine.device != x.device:
            engine.ensure(x.device)          # (first use / device move; engine.train_step runs the full check)
        self._ensure_moments(engine)
        g = self.param_groups[0]
        self._t += 1
        pred, loss = engine.train_step(x, target, self._exp_avg, self._exp_avg_sq, float(g["lr"]),
                                       g["betas"], g["eps"], max_norm, self._t, self._stats, loss_out=loss_out)
        self._sync_step_state(engine)
        for (_, p, _, _), view in zip(engine._named_params(validate=False), engine.grad_views()):
            if p.grad is None:
                p.grad = view
        return pred, loss

    @property
    def last_grad_norm_stats(self):
        """Device tensor [total_norm, clip_coef] of the last fused step."""
        return self._stats


Adam._step_with_hooks = torch.optim.Optimizer.profile_hook_step(Adam._step_impl)


@torch.no_grad()
def clip_grad_norm_(module_or_parameters, max_norm, module=None):
    """nn.utils.clip_grad_norm_(bilinear.parameters(), max_norm=1) of
    train_bilinear.py:81 on the flat gradient arena: one reduction + one scaling
    pass.  Accepts the BilinearUnit itself, or its parameters plus ``module=``.
    Returns the total norm as a device tensor (no host synchronisation)."""
    import ctypes

    from . import _native as N
    if module is None:
        module = module_or_parameters
    engine = _engine_of(module)
    if engine.params is None:
        raise RuntimeError("no gradients: run a forward/backward first")
    if not engine.grads_in_arena():
        for (_, p, off, shape), ptr in zip(engine._named_params(), engine.grad_ptrs()):
            g = p.grad
            if g is None:
                raise RuntimeError("clip_grad_norm_: a parameter has no gradient")
            if g.data_ptr() != ptr:
                view = engine.grad_view(off, shape)
                view.copy_(g)
                p.grad = view
    stats = torch.empty(2, dtype=torch.float32, device=engine.device)
    sc = engine.scratch()
    N.check(N.lib().blh_clip_grad_norm(engine._stream(), N.ptr(engine.grads), engine.layout.total,
                                       float(max_norm), N.ptr(sc), sc.numel(), N.ptr(stats)),
            "blh_clip_grad_norm")
    return stats[0]
