"""Host-side owner of the device state the native kernels work on.

One flat fp32 arena each for parameters and gradients (plus Adam's two, owned by
``bilinear_amd.optim.Adam``), BatchNorm running statistics, and a per-batch-size
workspace.  The ``nn.Parameter`` / buffer objects of ``BilinearUnit`` are *views*
into these arenas, so ``state_dict`` / ``load_state_dict`` / ``.apply(init)`` of
the reference surface (/root/reference/model/bilinear.py:58-92,
train_bilinear.py:92-104) keep working while the kernels see contiguous memory.

PyTorch is plumbing here: it allocates device memory and provides the stream.
All arithmetic is done by libbilinear_hip.so through ``_native``.
"""
from __future__ import annotations

import ctypes

import torch

from . import _native as N
from . import ops as _ops  # noqa: F401  (registers torch.ops.bilinear_hip.*)

IN_FEATURES = 32    # 2 * 16 joints, /root/reference/model/bilinear.py:20-22
OUT_FEATURES = 48   # 3 * 16 joints, /root/reference/model/bilinear.py:29


def _require_hip(device):
    if device.type != "cuda":
        raise RuntimeError(
            "bilinear_amd runs only on a HIP device (MI355X); got a tensor on '%s'. "
            "There is no CPU path in the product." % device)


class ArenaLayout:
    """Names / offsets / shapes of the flat parameter arena, as the library lays it out."""

    def __init__(self, num_blocks, width, gemm_dtype=0):
        lib = N.lib()
        self.desc = N.ModelDesc(num_blocks, width, IN_FEATURES, OUT_FEATURES, int(gemm_dtype))
        total = lib.blh_param_arena_floats(ctypes.byref(self.desc))
        if total < 0:
            N.check(int(total), "blh_param_arena_floats(num_blocks=%d, width=%d)" % (num_blocks, width))
        self.total = int(total)
        self.num_heavy = int(lib.blh_num_heavy(ctypes.byref(self.desc)))
        self.entries = []          # (name, offset, shape)
        n = lib.blh_num_param_tensors(ctypes.byref(self.desc))
        buf = ctypes.create_string_buffer(64)
        off, rows, cols = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        for i in range(n):
            N.check(lib.blh_param_tensor_info(ctypes.byref(self.desc), i, buf, 64,
                                              ctypes.byref(off), ctypes.byref(rows),
                                              ctypes.byref(cols)), "blh_param_tensor_info")
            name = buf.value.decode()
            shape = (rows.value, cols.value) if cols.value > 1 else (rows.value,)
            self.entries.append((name, off.value, shape))
        self.bn_floats = int(lib.blh_bn_running_floats(ctypes.byref(self.desc)))

    def workspace_bytes(self, batch):
        b = N.lib().blh_workspace_bytes(ctypes.byref(self.desc), batch)
        if b < 0:
            N.check(int(b), "blh_workspace_bytes")
        return int(b)


class Engine:
    """Device arenas + calls into the native library for one BilinearUnit."""

    def __init__(self, module, num_blocks, width, gemm_dtype=0):
        self.layout = ArenaLayout(num_blocks, width, gemm_dtype)
        self.module = module
        self.width = width
        self.device = None
        self.params = None
        self.grads = None
        self.bn_running = None
        self.bn_nbt = None
        self._workspace = None
        self._scratch = None
        self.masks = None              # explicit keep-masks (parity tests), uint8 [nh,B,W]
        self.seed = int(torch.initial_seed()) & ((1 << 63) - 1)
        self.rng_step = 0
        self.row_offset = 0
        self._saved_batch = None       # batch of the forward whose state sits in the workspace
        self._saved_drop = None
        self._grad_ready_cb = None
        self.grad_ready_hook = None    # callable(offset, count): data-parallel bucket hook
        self.ctx = None                # N.Context on self.device (side stream, events, options)
        self.generation = 0            # counts train-mode forwards (guards backward, see _LifterFunction)
        self.shadow_epoch = 0          # counts invalidations of the persistent bf16 weight image
        self._param_versions = None    # sum of the Parameters' autograd version counters at the last fused step
        self._grad_ptr_cache = None
        self._grad_view_cache = None
        self._np_cache = None          # (_named_params list, module links, parameter links, BatchNorm modules)
        self._packed_cache = None      # (arena pointers, [(tensor owner, attribute or None, expected address)])

    # ---------------------------------------------------------------- arenas --
    def _named_params(self, validate=True):
        """[(name, Parameter, arena offset, shape)] in arena order.  Walking ``named_parameters()`` costs ~25 us
        and the drop-in step asks seven times per step (at the reference's batch of 64 the five-call step is
        host-bound), so the list is cached together with the module-tree links it was resolved through; the links
        (~50 dict lookups) are re-checked on every validating call, so replacing a submodule or a Parameter object is
        seen.  ``validate=False``: the callers that run BEHIND a forward of the same step (backward, clip_grad_norm_,
        optimizer.step — the forward validated) take the cached list as it is."""
        c = self._np_cache
        if c is not None and not validate:
            return c[0]
        if c is not None:
            ok = True
            for parent, key, child in c[1]:
                if parent._modules.get(key) is not child:
                    ok = False
                    break
            if ok:
                for owner, attr, par in c[2]:
                    if owner._parameters.get(attr) is not par:
                        ok = False
                        break
            if ok:
                return c[0]
        entries, mod_links, par_links, seen = [], [], [], set()
        for name, off, shape in self.layout.entries:
            parts = name.split(".")
            parent = self.module
            for key in parts[:-1]:
                child = parent._modules[key]
                if (id(parent), key) not in seen:
                    seen.add((id(parent), key))
                    mod_links.append((parent, key, child))
                parent = child
            par = parent._parameters[parts[-1]]
            par_links.append((parent, parts[-1], par))
            entries.append((name, par, off, shape))
        bns = [self.module._modules["encode"]._modules["1"]]
        for pair in self.module._modules["bilinear"]._modules.values():
            bns += [pair._modules["0"]._modules["1"], pair._modules["1"]._modules["1"]]
        self._np_cache = (entries, mod_links, par_links, bns)
        self._packed_cache = None
        return entries

    def _bn_modules(self):
        self._named_params()               # (validates / rebuilds the cache the BatchNorm list lives in)
        return self._np_cache[3]

    def is_packed(self, device, validate=True):
        """Do every Parameter and BatchNorm buffer still live in the arenas?  (A device address is unique to the device:
        comparing addresses covers the device check.)  The expected addresses are cached per arena: the per-call work is
        one data_ptr() per tensor — this runs in every forward, and the five-call step at batch 64 is host-bound."""
        if self.params is None or self.device != device:
            return False
        named = self._named_params(validate)
        key = (self.params.data_ptr(), self.bn_running.data_ptr(), self.bn_nbt.data_ptr())
        c = self._packed_cache
        if c is None or c[0] != key:
            W = self.width
            par = [(p, key[0] + 4 * off) for _, p, off, _ in named]
            buf = []
            for i, bn in enumerate(self._bn_modules()):
                buf += [(bn._buffers, "running_mean", key[1] + 4 * (2 * i) * W),
                        (bn._buffers, "running_var", key[1] + 4 * (2 * i + 1) * W),
                        (bn._buffers, "num_batches_tracked", key[2] + 8 * i)]
            c = self._packed_cache = (key, par, buf)
        for p, want in c[1]:
            if p.data_ptr() != want:
                return False
        for owner, name, want in c[2]:
            t = owner.get(name)
            if t is None or t.data_ptr() != want:
                return False
        return True

    def pack(self, device):
        """(Re)build the arenas on ``device`` from the module's current tensors and
        re-point every Parameter / BN buffer at its arena slot."""
        _require_hip(device)
        lay = self.layout
        params = torch.zeros(lay.total, dtype=torch.float32, device=device)
        grads = torch.zeros(lay.total, dtype=torch.float32, device=device)
        for _, p, off, shape in self._named_params():
            if tuple(p.shape) != tuple(shape):
                raise RuntimeError("parameter shape %s does not match arena slot %s"
                                   % (tuple(p.shape), tuple(shape)))
            slot = params[off:off + p.numel()].view(shape)
            slot.copy_(p.data.to(device=device, dtype=torch.float32))
            p.data = slot
            p.grad = None
        W = self.width
        running = torch.zeros(lay.num_heavy, 2, W, dtype=torch.float32, device=device)
        nbt = torch.zeros(lay.num_heavy, dtype=torch.int64, device=device)
        for i, bn in enumerate(self._bn_modules()):
            running[i, 0].copy_(bn.running_mean.to(device))
            running[i, 1].copy_(bn.running_var.to(device))
            nbt[i].copy_(bn.num_batches_tracked.to(device))
            bn._buffers["running_mean"] = running[i, 0]
            bn._buffers["running_var"] = running[i, 1]
            bn._buffers["num_batches_tracked"] = nbt[i]
        self.params, self.grads, self.bn_running, self.bn_nbt = params, grads, running, nbt
        if self.ctx is None or self.ctx.device != device:
            old = self.ctx
            self.ctx = N.Context(device)
            if old is None and self.layout.desc.gemm_dtype == 4:
                # bf16 storage: the fused step's Adam kernel keeps the bf16 weight image up to date and the next step
                # skips its re-cast of the arena (r06: -0.6 % at configs[2], -0.7 % at configs[4]'s shape,
                # profiles/r06_shadow_ab.txt).  Guarded by the Parameters' version counters, see train_step.
                self.ctx.set_option(N.OPT_PERSISTENT_SHADOW, 1)
            if old is not None:    # keep the options across a device move
                for opt in (N.OPT_TWO_STREAM, N.OPT_LATE_FORK, N.OPT_PERSISTENT_SHADOW,
                            N.OPT_SMALL_STEP):
                    self.ctx.set_option(opt, old.get_option(opt))
        self.device = device
        self._workspace = None
        self._saved_batch = None
        self.invalidate_shadow()

    def ensure(self, device, validate=True):
        _require_hip(device)
        if not self.is_packed(device, validate):
            self.pack(device)

    def grad_view(self, off, shape):
        # one as_strided instead of slice + view (this runs 22 times per step on the drop-in path)
        if len(shape) == 2:
            return self.grads.as_strided(shape, (shape[1], 1), off)
        return self.grads.as_strided(shape, (1,), off)

    def grad_views(self):
        """The arena slots as tensors, in _named_params() order — cached per gradient arena, so that handing a
        Parameter its ``.grad`` costs an attribute store instead of a new view (22 of them per backward)."""
        c = self._grad_view_cache
        if c is None or c[0] != self.grads.data_ptr():
            c = (self.grads.data_ptr(), [self.grad_view(off, shape) for _, off, shape in self.layout.entries])
            self._grad_view_cache = c
        return c[1]

    def grads_in_arena(self):
        """True when every Parameter's ``.grad`` IS the cached view of its arena slot (what backward hands out): the
        usual state between ``loss.backward()`` and ``optimizer.step()``; then nothing has to be gathered."""
        views = self.grad_views()
        for (_, p, _, _), v in zip(self._named_params(validate=False), views):
            if p.grad is not v:
                return False
        return True

    def grad_ptrs(self):
        """Device addresses of the arena slots, in _named_params() order (cached per gradient arena)."""
        c = self._grad_ptr_cache
        if c is None or c[0] != self.grads.data_ptr():
            base = self.grads.data_ptr()
            c = (base, [base + 4 * off for _, off, _ in self.layout.entries])
            self._grad_ptr_cache = c
        return c[1]

    def workspace(self, batch):
        need = self.layout.workspace_bytes(batch)
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = None
            self._workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._saved_batch = None
        return self._workspace

    def scratch(self):
        if self._scratch is None or self._scratch.device != self.device:
            self._scratch = torch.empty(1 << 16, dtype=torch.uint8, device=self.device)
        return self._scratch

    @staticmethod
    def _stream():
        return N.current_stream()

    _tuned = {}        # (device index, compute-stream handle) -> side-stream generation it was tuned against

    def _tune_streams(self):
        """Once per compute stream (and again whenever the library's side stream was replaced): let the
        library check that this stream and its side stream run beside each other, and pick another side
        stream if they do not (blh_tune_streams: a 2 ms probe).  On this stack a bad pair makes the
        two-stream backward 2x slower, and which pairs are bad depends on what the process created in
        which order — e.g. an RCCL communicator between the two streams (profiles/r04_dp_setup_order.md).
        BLH_NO_STREAM_TUNE=1 switches it off."""
        lib = N.lib()
        dev_index = torch._C._cuda_getDevice()
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        if raw is not None:          # (the fast check: no Stream object; this runs in every backward)
            if Engine._tuned.get((dev_index, raw(dev_index))) == lib.blh_side_stream_generation():
                return
        st = torch.cuda.current_stream()
        key = (st.device.index, st.cuda_stream)
        gen = lib.blh_side_stream_generation()
        if Engine._tuned.get(key) == gen:
            return
        import os
        if os.environ.get("BLH_NO_STREAM_TUNE") == "1" or torch.cuda.is_current_stream_capturing() \
                or not self.ctx.get_option(N.OPT_TWO_STREAM):
            return
        rep = (ctypes.c_float * 4)()
        N.check(lib.blh_tune_streams(ctypes.c_void_p(st.cuda_stream), 3, rep), "blh_tune_streams")
        self.stream_tune_report = tuple(rep)
        Engine._tuned[key] = lib.blh_side_stream_generation()
        alone, _, kept, tried = self.stream_tune_report
        if tried >= 0 and alone > 0 and kept > 2.5 * alone and not Engine._tune_warned:
            import warnings
            Engine._tune_warned = True
            warnings.warn("bilinear_amd: the compute stream and the library's side stream do not run beside each other "
                          "(probe %.2fx its solo time after %d replacement candidates; a good pair is ~1.7x): the "
                          "two-stream backward will be up to 2x slower.  Engine.set_two_stream(False), or create the "
                          "model after the process group / on another stream and call Engine.retune_streams()."
                          % (kept / alone, int(tried)), RuntimeWarning, stacklevel=3)

    _tune_warned = False

    @staticmethod
    def retune_streams():
        """Forget which compute streams were probed (the cache is keyed by the raw stream handle, which a destroyed
        and re-created stream can reuse): the next two-stream call of every engine probes again."""
        Engine._tuned.clear()

    def _momentum(self):
        m = self.module.encode[1].momentum
        return -1.0 if m is None else float(m)

    # --------------------------------------------------------------- dropout --
    def set_dropout_masks(self, masks):
        """Parity-test hook: replay explicit keep-masks (list of [B,W] arrays/tensors,
        one per heavy_linear) instead of the Philox stream.  ``None`` restores Philox."""
        if masks is None:
            self.masks = None
            return
        ms = [torch.as_tensor(m).to(torch.uint8) for m in masks]
        if len(ms) != self.layout.num_heavy:
            raise RuntimeError("need %d masks" % self.layout.num_heavy)
        self.masks = torch.stack(ms).contiguous()

    def _op_args(self):
        """(ctx, num_blocks, width, gemm_dtype) of the torch.ops.bilinear_hip.* schemas."""
        d = self.layout.desc
        return (int(self.ctx.handle.value), int(d.num_blocks), int(d.width), int(d.gemm_dtype))

    def _drop_struct(self, batch):
        if self.masks is not None:
            if self.masks.device != self.device:
                self.masks = self.masks.to(self.device)
            if tuple(self.masks.shape) != (self.layout.num_heavy, batch, self.width):
                raise RuntimeError("dropout masks have shape %s, expected %s" % (
                    tuple(self.masks.shape), (self.layout.num_heavy, batch, self.width)))
            return N.Dropout(self.masks.data_ptr(), 0, 0, 0, 0, 0)
        return N.Dropout(None, self.seed, self.rng_step, self.row_offset, 0, 0)

    # ---------------------------------------------------------------- compute --
    def _check_input(self, x):
        _require_hip(x.device)
        if x.dim() != 2 or x.shape[1] != IN_FEATURES:
            raise RuntimeError("expected input of shape [B, %d], got %s" % (IN_FEATURES, tuple(x.shape)))
        if x.dtype != torch.float32:
            raise RuntimeError("expected a float32 input, got %s" % x.dtype)
        return x.contiguous()

    def _native(self, name, *args):
        """One native entry point: the registered operator while torch.compile traces (that is what it can trace),
        the operator's implementation directly in eager mode (less host time per call).  Every tensor argument
        must be on this engine's device: a host pointer handed to the library would fault, so they are checked here
        (the operator dispatch would have refused them by itself)."""
        if torch.compiler.is_compiling():
            return getattr(torch.ops.bilinear_hip, name)(*args)
        dev = self.params.device          # (a tensor's device always carries its index)
        for a in args:
            if isinstance(a, torch.Tensor) and a.device != dev:
                raise RuntimeError("bilinear_hip::%s: a tensor on '%s' where '%s' is expected (there is no CPU "
                                   "implementation)" % (name, a.device, dev))
        return _ops.IMPLS[name](*args)

    def _sync_callback(self, ws, all_reduce_sum, errors):
        """ctypes callback for the SyncBN variants: wraps the exchange buffer (which lives in
        the workspace) as a tensor and hands it to ``all_reduce_sum`` (stream-ordered)."""
        base = ws.data_ptr()

        def _cb(user, ptr, count, dtype):
            if errors:
                return
            try:
                dt = torch.float64 if dtype == 1 else torch.float32
                nbytes = int(count) * (8 if dtype == 1 else 4)
                off = int(ptr) - base
                all_reduce_sum(ws[off:off + nbytes].view(dt))
            except BaseException as exc:   # noqa: BLE001  (must not unwind through C frames)
                errors.append(exc)
        return N.SyncFn(_cb)

    def forward_train(self, x, sync=None, global_batch=None, validated=False):
        """``sync`` (callable(tensor) -> in-place SUM all-reduce) selects SyncBN: batch
        statistics over ``global_batch`` rows across ranks.  ``validated``: the caller has just walked
        ``_named_params()`` (BilinearUnit.forward): the module-tree links are not re-checked."""
        x = self._check_input(x)
        self.ensure(x.device, validate=not validated)
        batch = x.shape[0]
        if batch < 2:
            raise ValueError("Expected more than 1 value per channel when training, got input size %s"
                             % (tuple(x.shape),))
        ws = self.workspace(batch)
        drop = self._drop_struct(batch)
        if sync is None:
            pred = self._native(
                "forward_train", x, self.params, self.bn_running, self.bn_nbt, ws, self.masks, *self._op_args(),
                self.seed, self.rng_step, self.row_offset, self._momentum())
        else:
            pred = torch.empty(batch, OUT_FEATURES, dtype=torch.float32, device=x.device)
            errors = []
            cb = self._sync_callback(ws, sync, errors)
            N.check(N.lib().blh_forward_train_sync(
                self.ctx.handle, ctypes.byref(self.layout.desc), self._stream(), N.ptr(self.params),
                N.ptr(self.bn_running), N.ptr(self.bn_nbt), N.ptr(x), ctypes.byref(drop),
                self._momentum(), N.ptr(ws), ws.numel(), N.ptr(pred), batch, int(global_batch),
                cb, None), "blh_forward_train_sync")
            if errors:
                raise errors[0]
        self._saved_batch = batch
        self._saved_drop = drop
        self.generation += 1
        if self.masks is None:
            self.rng_step += 1
        return pred

    def forward_train_autograd(self, x):
        """Train-mode forward as the DIFFERENTIABLE custom operator ``torch.ops.bilinear_hip.lifter_train``
        (torch.library.register_autograd, bilinear_amd/ops.py): ``loss.backward()`` of
        /root/reference/train_bilinear.py:79 reaches blh_backward through the operator's registered formula.  The
        operator is functional — it returns the activations it saved and the updated BatchNorm statistics instead
        of writing them behind the schema's back — so the two copy_ calls below ARE the buffer updates, visible to a
        tracing compiler; each Parameter's ``.grad`` becomes a view of the one gradient tensor the formula returns
        (arena layout: Adam.step / clip_grad_norm_ move it into the arena)."""
        x = self._check_input(x)
        self.ensure(x.device)
        batch = x.shape[0]
        if batch < 2:
            raise ValueError("Expected more than 1 value per channel when training, got input size %s"
                             % (tuple(x.shape),))
        named = self._named_params()
        # explicit dropout masks: moved to this device and shape-checked HERE — the operator's Python implementation
        # hands masks.data_ptr() to the kernels as it is (a host pointer or a mask built for another batch would be a
        # GPU memory fault, not an exception)
        self._drop_struct(batch)
        _ops.ENGINES[int(self.ctx.handle.value)] = self
        pred, _saved, new_running, new_nbt = torch.ops.bilinear_hip.lifter_train(
            x, [p for _, p, _, _ in named], self.params, self.bn_running, self.bn_nbt, self.masks,
            *self._op_args(), self.seed, self.rng_step, self.row_offset, self._momentum(),
            [int(off) for _, _, off, _ in named], self.layout.workspace_bytes(batch))
        self.bn_running.copy_(new_running.detach())
        self.bn_nbt.copy_(new_nbt.detach())
        self.generation += 1
        if self.masks is None:
            self.rng_step += 1
        return pred

    def forward_train_loss(self, x, target, sync=None, global_batch=None):
        """Train-mode forward + nn.MSELoss (train_bilinear.py:76,78) in one enqueue, as the fused
        step runs them: returns (pred, loss); the loss gradient stays in the workspace and the
        next ``backward(x, None, ...)`` picks it up (blh_forward_train_loss)."""
        x = self._check_input(x)
        self.ensure(x.device)
        batch = x.shape[0]
        if batch < 2:
            raise ValueError("Expected more than 1 value per channel when training, got input size %s"
                             % (tuple(x.shape),))
        target = target.contiguous()
        if tuple(target.shape) != (batch, OUT_FEATURES) or target.dtype != torch.float32:
            raise RuntimeError("bad target: %s %s" % (tuple(target.shape), target.dtype))
        ws = self.workspace(batch)
        drop = self._drop_struct(batch)
        pred = torch.empty(batch, OUT_FEATURES, dtype=torch.float32, device=x.device)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        lib = N.lib()
        if sync is None:
            N.check(lib.blh_forward_train_loss(
                self.ctx.handle, ctypes.byref(self.layout.desc), self._stream(), N.ptr(self.params),
                N.ptr(self.bn_running), N.ptr(self.bn_nbt), N.ptr(x), N.ptr(target), ctypes.byref(drop),
                self._momentum(), N.ptr(ws), ws.numel(), N.ptr(pred), N.ptr(loss), batch),
                "blh_forward_train_loss")
        else:
            errors = []
            cb = self._sync_callback(ws, sync, errors)
            N.check(lib.blh_forward_train_loss_sync(
                self.ctx.handle, ctypes.byref(self.layout.desc), self._stream(), N.ptr(self.params),
                N.ptr(self.bn_running), N.ptr(self.bn_nbt), N.ptr(x), N.ptr(target), ctypes.byref(drop),
                self._momentum(), N.ptr(ws), ws.numel(), N.ptr(pred), N.ptr(loss), batch,
                int(global_batch), cb, None), "blh_forward_train_loss_sync")
            if errors:
                raise errors[0]
        self._saved_batch = batch
        self._saved_drop = drop
        self.generation += 1
        if self.masks is None:
            self.rng_step += 1
        return pred, loss

    def forward_eval(self, x):
        x = self._check_input(x)
        self.ensure(x.device)
        batch = x.shape[0]
        ws = self.workspace(batch)
        self._saved_batch = None       # eval overwrites the saved activations
        return self._native("eval_fwd", x, self.params, self.bn_running, ws, *self._op_args())

    def set_persistent_shadow(self, enabled):
        """gemm_dtype "bf16s" only (BLH_OPT_PERSISTENT_SHADOW; ON by default in this host layer since round 6, off in
        the C ABI): the fused train step's Adam kernel also writes the bf16 image of the updated weights (and the
        decode weight's K-major image of the one-pass decode) and the next fused step skips re-casting the whole fp32
        arena.  Between two ``train_step`` calls the parameters must be written by nothing the engine cannot see:
        checkpoint loads, ``.apply(init)``, ``optimizer.step()`` and moving the module drop the image explicitly, and
        ANY in-place operation on a Parameter (``p.mul_()``, ``p.copy_()``, ``nn.init.*``) is caught through the
        Parameters' autograd version counters in front of every fused step.  What is invisible to both — edits
        through ``p.data`` or raw pointers — must be followed by ``invalidate_shadow()`` (or switch the image off
        here)."""
        if self.ctx is None:
            raise RuntimeError("the engine is not on a device yet")
        self.ctx.set_option(N.OPT_PERSISTENT_SHADOW, 1 if enabled else 0)

    def invalidate_shadow(self):
        """Drop the bf16 weight image kept under ``set_persistent_shadow``: the next fused step
        re-casts the arena."""
        self.shadow_epoch = getattr(self, "shadow_epoch", 0) + 1   # (captured steps compare it)
        if self.ctx is not None:
            self.ctx.set_option(N.OPT_PERSISTENT_SHADOW, self.ctx.get_option(N.OPT_PERSISTENT_SHADOW))

    def clip_adam_bf16(self, grads_bf16, grad_scale, exp_avg, exp_avg_sq, lr, betas, eps, max_norm, step,
                       stats=None):
        """clip + Adam reading the gradient from the bf16 buckets of the compressed data-parallel
        exchange (times ``grad_scale``); the fp32 gradient arena receives the clipped values."""
        hyper = N.AdamHyper(lr, betas[0], betas[1], eps, 0.0 if max_norm is None else max_norm, step, 0)
        sc = self.scratch()
        N.check(N.lib().blh_clip_adam_step_bf16(
            self._stream(), N.ptr(self.params), N.ptr(grads_bf16), float(grad_scale), N.ptr(self.grads),
            N.ptr(exp_avg), N.ptr(exp_avg_sq), self.layout.total, ctypes.byref(hyper), N.ptr(sc),
            sc.numel(), N.ptr(stats)), "blh_clip_adam_step_bf16")
        self.invalidate_shadow()          # the parameters changed behind any persistent bf16 image

    def set_two_stream(self, enabled):
        """A/B switch of the two-stream backward (bit-identical results either way)."""
        if self.ctx is None:
            raise RuntimeError("the engine is not on a device yet")
        self.ctx.set_option(N.OPT_TWO_STREAM, 1 if enabled else 0)

    def set_small_step(self, mode):
        """The small-batch kernels (at most 384 rows, fp32, BLH_OPT_SMALL_STEP): True / 1 / "staged" (default) = one
        launch per stage; False / 0 / "off" = the multi-launch path every other batch size takes."""
        if self.ctx is None:
            raise RuntimeError("the engine is not on a device yet")
        value = {"staged": 1, "off": 0}.get(mode, mode)
        self.ctx.set_option(N.OPT_SMALL_STEP, int(value))

    def backward(self, x, dpred, on_ready=None, sync=None, global_batch=None, generation=None):
        """Gradients of every parameter into the grad arena (overwritten).  ``generation`` (the
        value of ``self.generation`` right after the forward this backward belongs to) guards
        the single workspace: activations, batch statistics and the dropout step saved there
        are those of the LAST train-mode forward only."""
        batch = x.shape[0]
        if generation is not None and generation != self.generation:
            raise RuntimeError(
                "backward of a forward whose saved activations were overwritten: BilinearUnit keeps "
                "the state of the most recent train-mode forward only (forward %d, latest %d); run "
                "backward before the next forward" % (generation, self.generation))
        if self._saved_batch != batch:
            raise RuntimeError("backward called without a matching train-mode forward "
                               "(the workspace holds the activations of the last forward only)")
        if dpred is not None:      # (None: the loss gradient forward_train_loss left in the workspace)
            dpred = dpred.contiguous()
            if tuple(dpred.shape) != (batch, OUT_FEATURES) or dpred.dtype != torch.float32:
                raise RuntimeError("bad output gradient: %s %s" % (tuple(dpred.shape), dpred.dtype))
        ws = self.workspace(batch)
        self._tune_streams()
        errors = []
        if on_ready is not None:
            # a reported range is complete on the library's side stream (weight-gradient GEMMs):
            # the hook runs with that stream current, so a collective launched from it is
            # ordered behind the range without stalling the main stream
            side = self.ctx.side_stream()
            side_stream = torch.cuda.ExternalStream(side, device=x.device) if side else None

            def _hook(user, off, cnt):
                # an exception must not unwind through the C frames of blh_backward:
                # remember it and re-raise once the call has returned
                if not errors:
                    try:
                        if side_stream is not None:
                            with torch.cuda.stream(side_stream):
                                on_ready(int(off), int(cnt))
                        else:
                            on_ready(int(off), int(cnt))
                    except BaseException as exc:   # noqa: BLE001
                        errors.append(exc)
            cb = N.GradReadyFn(_hook)
        else:
            cb = ctypes.cast(None, N.GradReadyFn)
        self._grad_ready_cb = cb       # keep alive during the call
        if sync is None and on_ready is None and dpred is not None:
            d = self._saved_drop
            self._native(
                "backward", x, dpred, self.params, ws, self.grads, self.masks if d.keep_mask else None,
                *self._op_args(), int(d.seed), int(d.step), int(d.row_offset))
        elif sync is None:
            N.check(N.lib().blh_backward(
                self.ctx.handle, ctypes.byref(self.layout.desc), self._stream(), N.ptr(self.params), N.ptr(x),
                ctypes.byref(self._saved_drop), N.ptr(ws), ws.numel(), N.ptr(dpred),
                N.ptr(self.grads), batch, cb, None), "blh_backward")
        else:
            scb = self._sync_callback(ws, sync, errors)
            N.check(N.lib().blh_backward_sync(
                self.ctx.handle, ctypes.byref(self.layout.desc), self._stream(), N.ptr(self.params), N.ptr(x),
                ctypes.byref(self._saved_drop), N.ptr(ws), ws.numel(), N.ptr(dpred),
                N.ptr(self.grads), batch, cb, None, int(global_batch), scb, None),
                "blh_backward_sync")
        self._saved_batch = None
        if errors:
            raise errors[0]

    def mse_loss_grad(self, pred, target, denominator=None, grad_scale=1.0):
        """(loss scalar tensor, dpred) of nn.MSELoss — train_bilinear.py:49,78."""
        batch, of = pred.shape
        if denominator is None:
            denominator = float(batch * of)
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        dpred = torch.empty_like(pred)
        sc = self.scratch()
        N.check(N.lib().blh_mse_loss_grad(
            self._stream(), N.ptr(pred.contiguous()), N.ptr(target.contiguous()), batch, of,
            denominator, grad_scale, N.ptr(loss), N.ptr(dpred), N.ptr(sc), sc.numel()),
            "blh_mse_loss_grad")
        return loss, dpred

    def clip_adam(self, exp_avg, exp_avg_sq, lr, betas, eps, max_norm, step, stats=None):
        hyper = N.AdamHyper(lr, betas[0], betas[1], eps, 0.0 if max_norm is None else max_norm, step, 0)
        sc = self.scratch()
        self.invalidate_shadow()
        N.check(N.lib().blh_clip_adam_step(
            self._stream(), N.ptr(self.params), N.ptr(self.grads), N.ptr(exp_avg),
            N.ptr(exp_avg_sq), self.layout.total, ctypes.byref(hyper), N.ptr(sc), sc.numel(),
            N.ptr(stats)), "blh_clip_adam_step")

    def _check_shadow_versions(self):
        """bf16 storage: a Parameter written in place since the last fused step (its autograd version counter moved)
        makes the bf16 weight image stale — drop it."""
        if self.layout.desc.gemm_dtype == 4:
            versions = sum(p._version for _, p, _, _ in self._named_params())
            if versions != self._param_versions:
                self._param_versions = versions
                self.invalidate_shadow()

    def train_step(self, x, target, exp_avg, exp_avg_sq, lr, betas, eps, max_norm, step, stats=None, loss_out=None):
        """Whole step body of train_bilinear.py:75-83 as one native enqueue.  ``loss_out``: a 0-dim float32 device
        tensor that receives the loss (e.g. a slot of a loss ring, bilinear_amd.loss_log) instead of a fresh one."""
        x = self._check_input(x)
        self.ensure(x.device)
        batch = x.shape[0]
        if batch < 2:
            raise ValueError("Expected more than 1 value per channel when training")
        target = target.contiguous()
        if tuple(target.shape) != (batch, OUT_FEATURES) or target.dtype != torch.float32:
            raise RuntimeError("bad target: %s %s" % (tuple(target.shape), target.dtype))
        ws = self.workspace(batch)
        self._drop_struct(batch)          # validates explicit masks (shape, device)
        self._tune_streams()
        self._check_shadow_versions()
        args = (x, target, self.params, self.grads, exp_avg, exp_avg_sq, self.bn_running, self.bn_nbt,
                ws, stats, self.masks, *self._op_args(), self.seed, self.rng_step, self.row_offset,
                self._momentum(), float(lr), float(betas[0]), float(betas[1]), float(eps),
                0.0 if max_norm is None else float(max_norm), int(step))
        if loss_out is not None and not torch.compiler.is_compiling():
            if loss_out.dim() != 0 or loss_out.dtype != torch.float32 or loss_out.device != x.device:
                raise RuntimeError("loss_out must be a 0-dim float32 tensor on the input's device")
            pred, loss = _ops.train_step_into(loss_out, *args)
        else:
            pred, loss = self._native("train_step", *args)
        self.generation += 1
        self._saved_batch = None
        if self.masks is None:
            self.rng_step += 1
        return pred, loss
