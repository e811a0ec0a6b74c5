"""Drop-in for the reference's ``model.bilinear`` (/root/reference/model/bilinear.py):
the same three exports — ``heavy_linear``, ``BilinearUnit``, ``load`` — with the
same signatures, submodule names and ``state_dict`` keys, but every forward /
backward runs on hand-written gfx950 kernels through libbilinear_hip.so.

Differences a caller can see:
  * tensors must live on a HIP device (no CPU path: a CPU input raises);
  * ``BilinearUnit(num_blocks=2, width=1024)`` is parameterised (the reference
    hard-codes (2, 1024), model/bilinear.py:22-29);
  * ``load`` returns ``bilinear_amd.optim.Adam`` (same interface and
    ``state_dict`` format as ``torch.optim.Adam``; fused kernel underneath).
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from ..engine import IN_FEATURES, OUT_FEATURES, Engine
from ..optim import Adam

__all__ = ["heavy_linear", "BilinearUnit", "Bilinear", "load"]

GEMM_DTYPES = {"fp32": 0, "bf16x3": 2, "fp16x2": 3, "bf16s": 4}   # (1, round 1's mixed mode, is gone)


class _HeavyStageFunction(torch.autograd.Function):
    """Autograd bridge of one stand-alone stage (blh_heavy_forward / blh_heavy_backward)."""

    @staticmethod
    def forward(ctx, x, stage, weight, bias, gamma, beta):
        import ctypes

        from .. import _native as N
        lin, bn = stage[0], stage[1]
        batch = x.shape[0]
        in_f, out_f = lin.in_features, lin.out_features
        need = N.lib().blh_heavy_workspace_bytes(batch, in_f, out_f)
        if need < 0:
            N.check(int(need), "blh_heavy_workspace_bytes")
        ws = torch.empty(int(need), dtype=torch.uint8, device=x.device)
        out = torch.empty(batch, out_f, dtype=torch.float32, device=x.device)
        training = bool(stage.training)
        # every stand-alone stage draws its own Philox stream (layer_base = its process-unique id)
        drop = N.Dropout(None, stage._seed, stage._rng_step, 0, stage._stage_id, 0)
        momentum = -1.0 if bn.momentum is None else float(bn.momentum)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        hctx = N.default_context(x.device)
        N.check(N.lib().blh_heavy_forward(
            hctx.handle, st, N.ptr(x), N.ptr(weight), N.ptr(bias), N.ptr(gamma), N.ptr(beta),
            N.ptr(bn.running_mean), N.ptr(bn.running_var), N.ptr(bn.num_batches_tracked),
            ctypes.byref(drop), momentum, int(training), 0, N.ptr(ws), ws.numel(), N.ptr(out),
            batch, in_f, out_f), "blh_heavy_forward")
        if training:
            stage._rng_step += 1
        ctx.saved = (x, weight, gamma, ws, drop, in_f, out_f, training)
        return out

    @staticmethod
    def backward(ctx, d_out):
        import ctypes

        from .. import _native as N
        x, weight, gamma, ws, drop, in_f, out_f, training = ctx.saved
        if not training:
            raise RuntimeError("backward through an eval-mode heavy_linear stage is not supported")
        batch = x.shape[0]
        d_out = d_out.contiguous()
        dw = torch.empty_like(weight)
        db = torch.empty(out_f, dtype=torch.float32, device=x.device)
        dg = torch.empty_like(db)
        dbeta = torch.empty_like(db)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        N.check(N.lib().blh_heavy_backward(
            N.default_context(x.device).handle, st, N.ptr(d_out), N.ptr(x), N.ptr(weight), N.ptr(gamma), ctypes.byref(drop), 0,
            N.ptr(ws), ws.numel(), N.ptr(dw), N.ptr(db), N.ptr(dg), N.ptr(dbeta), N.ptr(dx), batch,
            in_f, out_f), "blh_heavy_backward")
        return dx, None, dw, db, dg, dbeta


class _HeavyLinear(nn.Sequential):
    """Linear -> BatchNorm1d -> ReLU -> Dropout(0.5) as ONE fused stage.

    Inside ``BilinearUnit`` the children only hold parameters and buffers (the unit drives all
    stages natively, fused across stage boundaries).  Called on its own, the stage runs the
    same kernels through blh_heavy_forward / blh_heavy_backward (HIP device only)."""

    # Philox stream ids of stand-alone stages start above any BilinearUnit's stage indices
    # (<= 32), one per constructed stage: stacked stages of equal shape never share a mask
    _next_stage_id = [1000]

    def __init__(self, *mods):
        super().__init__(*mods)
        object.__setattr__(self, "_stage_id", _HeavyLinear._next_stage_id[0])
        _HeavyLinear._next_stage_id[0] += 1

    def forward(self, in_tensor):   # noqa: D401
        if in_tensor.device.type != "cuda":
            raise RuntimeError(
                "bilinear_amd.heavy_linear runs only on a HIP device (MI355X); input is on '%s' "
                "and there is no CPU/ATen fallback." % in_tensor.device)
        lin, bn = self[0], self[1]
        if in_tensor.dim() != 2 or in_tensor.shape[1] != lin.in_features or in_tensor.dtype != torch.float32:
            raise RuntimeError("expected a float32 input of shape [B, %d]" % lin.in_features)
        if self.training and in_tensor.shape[0] < 2:
            raise ValueError("Expected more than 1 value per channel when training")
        if not hasattr(self, "_rng_step"):
            object.__setattr__(self, "_rng_step", 0)
            object.__setattr__(self, "_seed", int(torch.initial_seed()) & ((1 << 63) - 1))
        x = in_tensor.contiguous()
        args = (x, self, lin.weight.contiguous(), lin.bias.contiguous(), bn.weight.contiguous(),
                bn.bias.contiguous())
        if torch.is_grad_enabled() and self.training:
            return _HeavyStageFunction.apply(*args)
        with torch.no_grad():
            return _HeavyStageFunction.apply(*args)


def heavy_linear(in_features, out_features, bias=True):
    """/root/reference/model/bilinear.py:7-13 — same container layout
    (index 0 Linear, 1 BatchNorm1d, 2 ReLU, 3 Dropout(p=0.5))."""
    if not bias:
        raise RuntimeError("bilinear_amd kernels assume Linear(bias=True), as the reference uses")
    return _HeavyLinear(
        nn.Linear(in_features, out_features, bias=True),
        nn.BatchNorm1d(out_features),
        nn.ReLU(),
        nn.Dropout(p=0.5),
    )


# Which autograd bridge the eager drop-in forward takes: "op" = the differentiable custom operator
# (torch.ops.bilinear_hip.lifter_train, the only form torch.compile can trace — it is always used while compiling),
# "function" (default) = the plain autograd.Function below: same kernels, bit-identical results, 70 us less host
# time per step (tools_dev/dropin_b64.py: the five-call step at batch 64 is host-bound).
EAGER_AUTOGRAD = "function"


class _LifterFunction(torch.autograd.Function):
    """Autograd bridge for the drop-in surface: ``loss.backward()``
    (train_bilinear.py:79) lands in Engine.backward, which fills the flat gradient
    arena; each Parameter's ``.grad`` becomes a view of its arena slot."""

    @staticmethod
    def forward(ctx, x, engine, *params):
        pred = engine.forward_train(x, validated=True)      # (BilinearUnit.forward walked _named_params() for *params)
        ctx.engine = engine
        ctx.generation = engine.generation
        ctx.x = x
        ctx.params = params
        return pred

    @staticmethod
    def backward(ctx, dpred):
        engine = ctx.engine
        accumulate = any(p.grad is not None for p in ctx.params)
        old = engine.grads.clone() if accumulate else None
        hook = engine.grad_ready_hook
        engine.backward(ctx.x, dpred, on_ready=hook, generation=ctx.generation)
        for (_, p, off, shape), view in zip(engine._named_params(validate=False), engine.grad_views()):
            if p.grad is None:
                p.grad = view
            elif p.grad.data_ptr() == view.data_ptr():
                view.add_(old[off:off + view.numel()].view(shape))
            else:
                p.grad.add_(view)
        return (None, None) + tuple(None for _ in ctx.params)


class BilinearUnit(nn.Module):
    """/root/reference/model/bilinear.py:16-55."""

    def __init__(self, num_blocks=2, width=1024, gemm_dtype="fp32"):
        super().__init__()
        self.num_blocks = int(num_blocks)
        self.width = int(width)
        if gemm_dtype not in GEMM_DTYPES:
            raise ValueError("gemm_dtype must be one of %s" % sorted(GEMM_DTYPES))
        # "fp32": exact fp32 MFMA, the reference's arithmetic.  "bf16s": bf16 STORAGE — activations,
        # gradients and a weight shadow are bf16 in HBM, fp32 master weights / BatchNorm statistics
        # / Adam (BASELINE configs 3-5).  "bf16x3" / "fp16x2": fp32 accuracy on the 16-bit matrix cores
        self.gemm_dtype = gemm_dtype
        self.encode = heavy_linear(in_features=IN_FEATURES, out_features=self.width)
        self.bilinear = nn.ModuleList([
            nn.Sequential(
                heavy_linear(in_features=self.width, out_features=self.width),
                heavy_linear(in_features=self.width, out_features=self.width),
            ) for _ in range(self.num_blocks)
        ])
        self.decode = nn.Linear(in_features=self.width, out_features=OUT_FEATURES, bias=True)
        # not a submodule / parameter: excluded from state_dict
        object.__setattr__(self, "_engine", None)

    # -- native engine ---------------------------------------------------------
    @property
    def engine(self):
        if self._engine is None:
            eng = Engine(self, self.num_blocks, self.width, GEMM_DTYPES[self.gemm_dtype])
            object.__setattr__(self, "_engine", eng)
        return self._engine

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_engine"] = None          # device arenas are rebuilt lazily
        return state

    def forward(self, in_tensor):
        """model/bilinear.py:31-41.  train mode: batch statistics + dropout;
        eval mode (valid_bilinear.py:31): running statistics, no dropout."""
        eng = self.engine
        if in_tensor.device.type != "cuda":
            raise RuntimeError(
                "bilinear_amd.BilinearUnit runs only on a HIP device (MI355X); input is on '%s' "
                "and there is no CPU fallback." % in_tensor.device)
        # (the engine's forward_* methods re-check that the parameters still live in the arenas: eng.ensure)
        if not self.training:
            return eng.forward_eval(in_tensor)
        if torch.is_grad_enabled():
            params = [p for _, p, _, _ in eng._named_params()]
            # the differentiable custom operator (torch.library.register_autograd) — unless gradients are being
            # accumulated into existing .grad tensors or a data-parallel bucket hook wants the ranges as they
            # complete: those need Python between the kernels (_LifterFunction)
            if eng.grad_ready_hook is None and all(p.grad is None for p in params) and (
                    EAGER_AUTOGRAD == "op" or torch.compiler.is_compiling()):
                return eng.forward_train_autograd(in_tensor)
            return _LifterFunction.apply(in_tensor, eng, *params)
        return eng.forward_train(in_tensor)

    # parameter writes that do not go through the fused step drop the bf16 weight image kept under
    # Engine.set_persistent_shadow (a no-op otherwise)
    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        if self._engine is not None:
            self._engine.invalidate_shadow()
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        if self._engine is not None:
            self._engine.invalidate_shadow()
        return out

    def apply(self, fn):
        out = super().apply(fn)
        if self._engine is not None:
            self._engine.invalidate_shadow()
        return out

    def reset_statistics(self):
        """model/bilinear.py:43-55: reset every BN's running stats and switch it to
        the cumulative moving average (momentum=None)."""
        for key in self.state_dict().keys():
            if "running_mean" in key:
                layer = self
                for part in key.split(".")[:-1]:
                    layer = layer[int(part)] if part.isdigit() else getattr(layer, part)
                layer.reset_running_stats()
                layer.momentum = None

    def train_step(self, optimizer, x, target, max_norm=1.0, loss_out=None):
        """Fast path for the step body of train_bilinear.py:75-83: zero_grad, forward,
        MSELoss, backward, clip_grad_norm_(max_norm), Adam.step as ONE native enqueue
        (no autograd graph, no host synchronisation).  Returns (prediction, loss); ``loss_out``
        (a 0-dim float32 device tensor, e.g. ``bilinear_amd.LossRing.slot()``) receives the loss."""
        return optimizer.fused_train_step(self, x, target, max_norm, loss_out=loss_out)


Bilinear = BilinearUnit   # BASELINE.json names the class `Bilinear`; the reference calls it BilinearUnit


def load(device, parameter_dir=None, learning_rate=1.0e-3, num_blocks=2, width=1024,
         gemm_dtype="fp32"):
    """/root/reference/model/bilinear.py:58-92 — build the module on ``device``,
    an Adam optimiser, and either restore the newest ``{epoch}.save`` checkpoint
    under ``parameter_dir`` or Kaiming-initialise every Linear weight.
    Returns (module, optimizer, step, epoch_to_load)."""
    bilinear = BilinearUnit(num_blocks=num_blocks, width=width, gemm_dtype=gemm_dtype).to(device)
    optimizer = Adam(bilinear.parameters(), lr=learning_rate, module=bilinear)
    step = 1

    epoch_to_load = 0
    if parameter_dir is not None:
        for _, _, files in os.walk(parameter_dir):
            for file in files:
                name, extension = file.split(".")     # "{epoch}.save"
                epoch_to_load = max(epoch_to_load, int(name))

    if epoch_to_load != 0:
        parameter_file = "{parameter_dir}/{epoch}.save".format(
            parameter_dir=parameter_dir, epoch=epoch_to_load)
        parameter = torch.load(parameter_file, map_location=device, weights_only=False)
        bilinear.load_state_dict(parameter["state"])
        optimizer.load_state_dict(parameter["optimizer"])
        step = parameter["step"]
    else:
        def weight_init(m):
            if isinstance(m, nn.Linear):
                nn.init.kaiming_normal_(m.weight)
        bilinear.apply(weight_init)

    return bilinear, optimizer, step, epoch_to_load
