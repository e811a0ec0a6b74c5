"""``torch.ops.bilinear_hip.*`` — the C ABI of libbilinear_hip.so registered as PyTorch-ROCm
custom operators (BASELINE.json north_star: "driven from Python through PyTorch-ROCm custom
ops").  Each operator is a thin schema over ONE entry point of include/bilinear_hip.h; tensors
supply device memory, the current HIP stream is taken from PyTorch, everything numerical happens
in the library.  The drop-in module (bilinear_amd/model/bilinear.py) and the engine call these
operators; the data-parallel paths that pass host callbacks (bucket hook, SyncBN exchange) go to
the C ABI directly, since a callback has no operator-schema type.

    pred = torch.ops.bilinear_hip.eval_fwd(x, params, bn_running, workspace, ctx, nb, W, dtype)
    pred = torch.ops.bilinear_hip.forward_train(x, params, bn_running, bn_nbt, workspace, masks, ...)
    pred, saved, running, nbt = torch.ops.bilinear_hip.lifter_train(x, param_views, params, ...)   # differentiable
    torch.ops.bilinear_hip.backward(x, dpred, params, workspace, grads, masks, ...)
    pred, loss = torch.ops.bilinear_hip.train_step(x, target, params, grads, exp_avg, ...)

Reference call-sites: BilinearUnit.forward (/root/reference/model/bilinear.py:31-41, eval:
valid_bilinear.py:31,52), loss.backward() (train_bilinear.py:79), the step body
(train_bilinear.py:75-83).  ``ctx`` is the integer value of a ``blh_context*`` (bilinear_amd.
_native.Context.handle.value).  There is no CPU implementation: calling an operator with CPU
tensors raises (no kernel registered for the CPU dispatch key).
"""
from __future__ import annotations

import ctypes

import torch

from . import _native as N

IN_FEATURES, OUT_FEATURES = 32, 48

_LIB = torch.library.Library("bilinear_hip", "DEF")

_LIB.define(
    "eval_fwd(Tensor x, Tensor params, Tensor bn_running, Tensor(a!) workspace, int ctx, "
    "int num_blocks, int width, int gemm_dtype) -> Tensor")
_LIB.define(
    "forward_train(Tensor x, Tensor params, Tensor(a!) bn_running, Tensor(b!) bn_nbt, "
    "Tensor(c!) workspace, Tensor? masks, int ctx, int num_blocks, int width, int gemm_dtype, "
    "int seed, int step, int row_offset, float momentum) -> Tensor")
# The differentiable form of forward_train (torch.library.register_autograd below).  torch.library registers autograd
# formulas for FUNCTIONAL schemas only, and a functional schema is a promise the compiler stack acts on (dead-code
# elimination, re-ordering, static-input copies under CUDA-graph modes), so this operator really is functional
# (round 5; round 4 declared the buffers the native call writes as plain inputs): it mutates NONE of its inputs.
# The native forward runs on a clone of the BatchNorm running statistics / counter and on a workspace the operator
# allocates, and all three are RETURNED — (pred, saved, new_running, new_nbt); the caller copies the statistics
# back into the module's buffers with ordinary copy_ calls (Engine.forward_train_autograd), which a traced graph
# sees as what they are.  ``param_views`` are the module's nn.Parameters — views of ``params`` (the arena) at
# ``offsets`` — and exist in the schema only so that autograd connects them to the output; the kernels read the
# arena.  The formula calls ``lifter_backward``: gradients come back as ONE fresh tensor in the arena's layout
# (the .grad of each Parameter is a view of it); it uses the saved workspace as scratch and says so (Tensor(a!)).
_LIB.define(
    "lifter_train(Tensor x, Tensor[] param_views, Tensor params, Tensor bn_running, Tensor bn_nbt, "
    "Tensor? masks, int ctx, int num_blocks, int width, int gemm_dtype, int seed, int step, int row_offset, "
    "float momentum, int[] offsets, int workspace_bytes) -> (Tensor, Tensor, Tensor, Tensor)")
_LIB.define(
    "lifter_backward(Tensor x, Tensor dpred, Tensor params, Tensor(a!) saved, Tensor? masks, int ctx, "
    "int num_blocks, int width, int gemm_dtype, int seed, int step, int row_offset) -> Tensor")
_LIB.define(
    "backward(Tensor x, Tensor dpred, Tensor params, Tensor(a!) workspace, Tensor(b!) grads, "
    "Tensor? masks, int ctx, int num_blocks, int width, int gemm_dtype, int seed, int step, "
    "int row_offset) -> ()")
_LIB.define(
    "train_step(Tensor x, Tensor target, Tensor(a!) params, Tensor(b!) grads, Tensor(c!) exp_avg, "
    "Tensor(d!) exp_avg_sq, Tensor(e!) bn_running, Tensor(f!) bn_nbt, Tensor(g!) workspace, "
    "Tensor(h!)? stats, Tensor? masks, int ctx, int num_blocks, int width, int gemm_dtype, "
    "int seed, int step, int row_offset, float momentum, float lr, float beta1, float beta2, "
    "float eps, float max_norm, int adam_step) -> (Tensor, Tensor)")


def _stream():
    return N.current_stream()


def _desc(num_blocks, width, gemm_dtype):
    return N.ModelDesc(int(num_blocks), int(width), IN_FEATURES, OUT_FEATURES, int(gemm_dtype))


def _drop(masks, seed, step, row_offset):
    if masks is not None:
        return N.Dropout(masks.data_ptr(), 0, 0, 0, 0, 0)
    return N.Dropout(None, int(seed), int(step), int(row_offset), 0, 0)


def _eval_fwd(x, params, bn_running, workspace, ctx, num_blocks, width, gemm_dtype):
    batch = x.shape[0]
    pred = torch.empty(batch, OUT_FEATURES, dtype=torch.float32, device=x.device)
    d = _desc(num_blocks, width, gemm_dtype)
    N.check(N.lib().blh_forward_eval(
        ctypes.c_void_p(ctx), ctypes.byref(d), _stream(), N.ptr(params), N.ptr(bn_running), N.ptr(x),
        N.ptr(workspace), workspace.numel(), N.ptr(pred), batch), "blh_forward_eval")
    return pred


def _forward_train(x, params, bn_running, bn_nbt, workspace, masks, ctx, num_blocks, width,
                   gemm_dtype, seed, step, row_offset, momentum):
    batch = x.shape[0]
    pred = torch.empty(batch, OUT_FEATURES, dtype=torch.float32, device=x.device)
    d = _desc(num_blocks, width, gemm_dtype)
    drop = _drop(masks, seed, step, row_offset)
    N.check(N.lib().blh_forward_train(
        ctypes.c_void_p(ctx), ctypes.byref(d), _stream(), N.ptr(params), N.ptr(bn_running),
        N.ptr(bn_nbt), N.ptr(x), ctypes.byref(drop), float(momentum), N.ptr(workspace),
        workspace.numel(), N.ptr(pred), batch), "blh_forward_train")
    return pred


def _lifter_train(x, param_views, params, bn_running, bn_nbt, masks, ctx, num_blocks, width, gemm_dtype, seed, step,
                  row_offset, momentum, offsets, workspace_bytes):
    new_running, new_nbt = bn_running.clone(), bn_nbt.clone()
    saved = torch.empty(int(workspace_bytes), dtype=torch.uint8, device=x.device)
    pred = _forward_train(x, params, new_running, new_nbt, saved, masks, ctx, num_blocks, width, gemm_dtype,
                          seed, step, row_offset, momentum)
    return pred, saved, new_running, new_nbt


def _lifter_backward(x, dpred, params, saved, masks, ctx, num_blocks, width, gemm_dtype, seed, step, row_offset):
    grads = torch.empty_like(params)
    _backward(x, dpred.contiguous(), params, saved, grads, masks, ctx, num_blocks, width, gemm_dtype, seed, step,
              row_offset)
    return grads


def _backward(x, dpred, params, workspace, grads, masks, ctx, num_blocks, width, gemm_dtype, seed,
              step, row_offset):
    batch = x.shape[0]
    d = _desc(num_blocks, width, gemm_dtype)
    drop = _drop(masks, seed, step, row_offset)
    N.check(N.lib().blh_backward(
        ctypes.c_void_p(ctx), ctypes.byref(d), _stream(), N.ptr(params), N.ptr(x),
        ctypes.byref(drop), N.ptr(workspace), workspace.numel(), N.ptr(dpred), N.ptr(grads), batch,
        ctypes.cast(None, N.GradReadyFn), None), "blh_backward")


def _train_step(x, target, params, grads, exp_avg, exp_avg_sq, bn_running, bn_nbt, workspace,
                stats, masks, ctx, num_blocks, width, gemm_dtype, seed, step, row_offset, momentum,
                lr, beta1, beta2, eps, max_norm, adam_step, loss_out=None):
    batch = x.shape[0]
    pred = torch.empty(batch, OUT_FEATURES, dtype=torch.float32, device=x.device)
    loss = torch.empty((), dtype=torch.float32, device=x.device) if loss_out is None else loss_out
    d = _desc(num_blocks, width, gemm_dtype)
    drop = _drop(masks, seed, step, row_offset)
    hyper = N.AdamHyper(lr, beta1, beta2, eps, max_norm, int(adam_step), 0)
    N.check(N.lib().blh_train_step(
        ctypes.c_void_p(ctx), ctypes.byref(d), _stream(), N.ptr(params), N.ptr(grads),
        N.ptr(exp_avg), N.ptr(exp_avg_sq), N.ptr(bn_running), N.ptr(bn_nbt), N.ptr(x),
        N.ptr(target), ctypes.byref(drop), float(momentum), ctypes.byref(hyper), N.ptr(workspace),
        workspace.numel(), N.ptr(pred), N.ptr(loss), N.ptr(stats), batch), "blh_train_step")
    return pred, loss


# HIP device only ("CUDA" is PyTorch-ROCm's dispatch key for HIP tensors): no CPU kernels exist
_LIB.impl("eval_fwd", _eval_fwd, "CUDA")
_LIB.impl("forward_train", _forward_train, "CUDA")
_LIB.impl("lifter_train", _lifter_train, "CUDA")
_LIB.impl("lifter_backward", _lifter_backward, "CUDA")
_LIB.impl("backward", _backward, "CUDA")
_LIB.impl("train_step", _train_step, "CUDA")


def _fake_pred(x, *args, **kwargs):
    return x.new_empty((x.shape[0], OUT_FEATURES))


def _fake_step(x, *args, **kwargs):
    return x.new_empty((x.shape[0], OUT_FEATURES)), x.new_empty(())


# shape functions (FakeTensor / torch.compile tracing; no arithmetic)
torch.library.register_fake("bilinear_hip::eval_fwd", _fake_pred, lib=_LIB)
torch.library.register_fake("bilinear_hip::forward_train", _fake_pred, lib=_LIB)


def _fake_lifter(x, param_views, params, bn_running, bn_nbt, masks, ctx, num_blocks, width, gemm_dtype, seed, step,
                 row_offset, momentum, offsets, workspace_bytes):
    return (x.new_empty((x.shape[0], OUT_FEATURES)), x.new_empty((workspace_bytes,), dtype=torch.uint8),
            torch.empty_like(bn_running), torch.empty_like(bn_nbt))


torch.library.register_fake("bilinear_hip::lifter_train", _fake_lifter, lib=_LIB)
torch.library.register_fake("bilinear_hip::lifter_backward", lambda x, dpred, params, *a, **k: torch.empty_like(params),
                            lib=_LIB)
torch.library.register_fake("bilinear_hip::backward", lambda *a, **k: None, lib=_LIB)
torch.library.register_fake("bilinear_hip::train_step", _fake_step, lib=_LIB)



# ---- autograd of lifter_train: loss.backward() of /root/reference/train_bilinear.py:79 -----------------------
import weakref  # noqa: E402

ENGINES = weakref.WeakValueDictionary()      # engines by context handle (Engine.forward_train_autograd registers)


def _lifter_setup(ctx, inputs, output):
    (x, param_views, params, bn_running, bn_nbt, masks, c, nb, w, dt, seed, step, row_offset, momentum, offsets,
     workspace_bytes) = inputs
    saved = output[1]
    # (only the prediction is differentiable: without this the statistics copied back into the module's buffers
    #  would chain every step's graph to the previous one)
    ctx.mark_non_differentiable(output[1], output[2], output[3])
    ctx.save_for_backward(x, params, saved, masks)
    ctx.ints = (c, nb, w, dt, seed, step, row_offset)
    ctx.slots = [(int(o), tuple(p.shape)) for o, p in zip(offsets, param_views)]


def _lifter_formula(ctx, dpred, dsaved, drunning, dnbt):
    x, params, saved, masks = ctx.saved_tensors
    grads = torch.ops.bilinear_hip.lifter_backward(x, dpred, params, saved, masks, *ctx.ints)
    views = [grads.as_strided(shape, (shape[1], 1) if len(shape) == 2 else (1,), o) for o, shape in ctx.slots]
    return (None, views) + (None,) * 14


torch.library.register_autograd("bilinear_hip::lifter_train", _lifter_formula, setup_context=_lifter_setup, lib=_LIB)

def train_step_into(loss_out, *args):
    """The eager fused step with the loss written into ``loss_out`` (a slot of a loss ring: no extra launch)."""
    return _train_step(*args, loss_out=loss_out)


OPS = ("eval_fwd", "forward_train", "lifter_train", "lifter_backward", "backward", "train_step")

# the same entry points without the operator dispatch (~20 us of host time per call): what the engine calls in eager
# mode once IT has checked that every tensor is on the HIP device (a host pointer handed to the library would fault)
IMPLS = {"eval_fwd": _eval_fwd, "forward_train": _forward_train, "backward": _backward, "train_step": _train_step}
