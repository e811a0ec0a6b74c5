"""Data parallelism for the training step (new work: the reference is single-device,
/root/reference/util/config.py:17).  One process per GPU; parameters and Adam state
are replicated; the batch is split by rows; the only exchange per step is the
all-reduce of the flat gradient arena (RCCL over xGMI when the process group is
``nccl``; ``gloo`` in the CPU tests).

Overlap: blh_backward reports, on the host, each contiguous gradient range as soon
as the kernels that produce it are enqueued (decode first, encode last).  The reducer
launches an asynchronous all-reduce per bucket right there; RCCL runs it on its own
stream behind an event on the compute stream, so the exchange of layer l overlaps the
backward GEMMs of layers < l.  ``finish()`` makes the compute stream wait for all
buckets and the (already averaged) gradients then feed the fused clip + Adam, which
every rank runs identically (no parameter broadcast).

The returned loss is the global batch's (a 1-float all-reduce in front of backward; every
rank computes the same clip coefficient from the identical averaged gradients, so the gradient
norm needs no exchange of its own).

BatchNorm statistics are per-rank by default (the usual DDP semantics).  ``sync_bn=True``
exchanges every stage's statistics across ranks (2W-element all-reduces, forward and
backward), which reproduces the reference's single-device result on the concatenated
batch exactly (SURVEY.md hazard H5) at the price of 2*(1+L) latency-bound collectives.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


_COMPUTE_STREAMS = {}     # device -> the high-priority compute stream of DataParallel.stream


def _row_stride(batch):
    """Global index of a rank's first row = rank * this: the per-rank batch, rounded up to the 32 rows the Philox
    generator's row blocks span (include/bilinear_hip.h: blh_dropout.row_offset must be a multiple of 32).  For the
    usual per-rank batches (multiples of 32) a row's mask does not depend on the number of ranks; a ragged per-rank
    batch — the reference loader's last batch split over the ranks — still gets distinct masks on every rank instead
    of a refused call."""
    return (int(batch) + 31) // 32 * 32


class GradBucketReducer:
    """Bucketed, overlapped all-reduce (mean) over a flat gradient tensor.

    ``on_ready(offset, count)`` is called in backward order with contiguous ranges;
    ranges are merged until a bucket holds at least ``bucket_floats`` elements."""

    def __init__(self, flat_grads, group=None, bucket_floats=None, force_collectives=False,
                 compress=None):
        """``compress="bf16"``: every bucket is cast to bf16 (round to nearest even, HIP kernel),
        averaged in bf16 and cast back — half the bytes on every xGMI link (configs 3-5)."""
        if compress not in (None, "bf16"):
            raise ValueError("compress must be None or 'bf16'")
        self.compress = compress
        self._half = None
        self.flat = flat_grads
        self.group = group
        # default: about four buckets per step, none under 1 Mi elements — a bucket per 1024x1024 stage
        # costs one trip back to Python and one RCCL launch per stage (nine at configs[2], seventeen at
        # configs[4]) for no more overlap: the first quarter of the arena is on the wire after a
        # quarter of backward either way, and xGMI rings are per-link bound, so fewer, larger
        # collectives are the cheaper ones
        self.bucket_floats = int(bucket_floats) if bucket_floats else max(1 << 20, flat_grads.numel() // 4)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # world 1 normally skips the exchange; ``force_collectives`` issues it anyway (a one-GPU
        # box can then run every RCCL call of the N-GPU step: ReduceOp.AVG, the side-stream hook,
        # work.wait() ordering)
        self.force = bool(force_collectives) and dist.is_initialized()
        self._pending = None        # (lo, hi) not yet launched
        self._works = []
        self.launched = []          # [(lo, hi)] of the last step, for tests

    def begin(self):
        self._pending = None
        self._works = []
        self.launched = []

    def _launch(self, lo, hi):
        self.launched.append((lo, hi))
        if self.world == 1 and not self.force:
            return
        view = self.flat[lo:hi]
        backend = dist.get_backend(self.group)
        if self.compress == "bf16":
            if self._half is None:
                self._half = torch.empty(self.flat.numel(), dtype=torch.bfloat16, device=self.flat.device)
            half = self._half[lo:hi]
            self._cast(view, half, to_half=True)
            op = dist.ReduceOp.AVG if backend == "nccl" else dist.ReduceOp.SUM
            work = dist.all_reduce(half, op=op, group=self.group, async_op=True)
            self._works.append((work, ("half", lo, hi, backend != "nccl")))
            return
        if backend == "nccl":
            work = dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            self._works.append((work, None))
        else:
            work = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._works.append((work, view))

    def on_ready(self, offset, count):
        lo, hi = int(offset), int(offset + count)
        if self._pending is None:
            self._pending = (lo, hi)
        else:
            plo, phi = self._pending
            if hi == plo:                       # backward walks the arena downwards
                self._pending = (lo, phi)
            elif lo == phi:
                self._pending = (plo, hi)
            else:                               # not adjacent: flush what we have
                self._launch(plo, phi)
                self._pending = (lo, hi)
        plo, phi = self._pending
        if phi - plo >= self.bucket_floats:
            self._launch(plo, phi)
            self._pending = None

    def reduce_scalars(self, tensor):
        """Mean over ranks of a small tensor (the step's loss), launched like a
        bucket: asynchronous, completed by finish()."""
        if self.world == 1 and not self.force:
            return
        backend = dist.get_backend(self.group)
        if backend == "nccl":
            work = dist.all_reduce(tensor, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            self._works.append((work, None))
        else:
            work = dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._works.append((work, tensor))

    def flush(self):
        """Launch the bucket still pending (the last ranges of backward)."""
        if self._pending is not None:
            self._launch(*self._pending)
            self._pending = None

    def finish(self, cast_back=True):
        """Wait for every launched collective.  Compressed buckets: ``cast_back=True`` writes the
        averaged bf16 values back into the fp32 arena; ``cast_back=False`` leaves them in
        ``self._half`` for an optimiser that reads bf16 gradients (Engine.clip_adam_bf16) and
        returns the factor still to be applied (1 / world when the backend summed)."""
        self.flush()
        scale = 1.0
        for work, view in self._works:
            work.wait()
            if isinstance(view, tuple):            # compressed bucket
                _, lo, hi, divide = view
                if cast_back:
                    self._cast(self.flat[lo:hi], self._half[lo:hi], to_half=False)
                    if divide:
                        self.flat[lo:hi].div_(self.world)
                elif divide:
                    scale = 1.0 / self.world
            elif view is not None:
                view.div_(self.world)
        self._works = []
        return scale

    @staticmethod
    def _cast(full, half, to_half):
        """fp32 <-> bf16 on the current stream: the library's cast kernels on a HIP device
        (bucket bounds are multiples of 64 elements), torch on the CPU (gloo tests)."""
        if full.is_cuda:
            import ctypes

            from . import _native as N
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            if to_half:
                N.check(N.lib().blh_cast_f32_to_bf16(st, N.ptr(full), N.ptr(half), full.numel()),
                        "blh_cast_f32_to_bf16")
            else:
                N.check(N.lib().blh_cast_bf16_to_f32(st, N.ptr(half), N.ptr(full), full.numel()),
                        "blh_cast_bf16_to_f32")
        elif to_half:
            half.copy_(full)
        else:
            full.copy_(half)


class DataParallel:
    """Data-parallel driver of BilinearUnit's fast path.

        dp = DataParallel(net, opt)            # after dist.init_process_group(...)
        pred, loss = dp.train_step(x_local, t_local)

    ``x_local`` is this rank's row slice of the global batch.  Dropout uses the
    global row index (rank * local_batch), so the Philox mask of a row does not
    depend on the number of GPUs.
    Order of set-up: any.  (Round 3 prescribed "process group first": with the module built before
    ``init_process_group("nccl", ...)`` the step took 2.3-3.0 instead of 1.1 ms.  The cause was the pair
    (compute stream, side stream) landing on hardware queues that do not run beside each other, which
    every set-up order draws anew; the engine now measures the pair in front of the first two-stream
    call on a compute stream and replaces the side stream of a bad pair — ``Engine._tune_streams``,
    profiles/r04_dp_setup_order.md: 1.105-1.108 ms in every order.)  See also ``stream``.
    """

    def __init__(self, module, optimizer, group=None, bucket_floats=None, max_norm=1.0,
                 sync_bn=False, force_collectives=False, compress=None, collectives="torch",
                 native_tail="producer"):
        """``collectives``: ``"torch"`` (default) — every bucket all-reduce is launched from blh_backward's hook
        through ``torch.distributed``; ``"native"`` (opt-in, HIP devices) — the LIBRARY owns an RCCL
        communicator of its own (``blh_comm``: created here from a unique id broadcast over ``group``) and the whole
        step is one call, ``blh_train_step_dp``: bucket all-reduces enqueued by backward itself behind the kernel that
        completes each bucket, norm + clip + Adam right behind the last one, one join (csrc/comm.hip,
        profiles/r06_dp_overhead.md).  Same buckets, same arithmetic: the two modes give bit-identical parameters.
        N > 1 ranks were never available to this repository — both modes are executed at world size 1 with every
        collective issued (``force_collectives``) and over gloo on the CPU (torch mode only).
        ``native_tail``: ``"producer"`` (the last bucket and the optimiser ride the stream that produced the last
        gradient) or ``"comm"`` (they run on the communicator's stream)."""
        if collectives not in ("torch", "native"):
            raise ValueError("collectives must be 'torch' or 'native'")
        if native_tail not in ("producer", "comm"):
            raise ValueError("native_tail must be 'producer' or 'comm'")
        if compress not in (None, "bf16"):
            raise ValueError("compress must be None or 'bf16'")
        self.collectives = collectives
        self.native_tail = native_tail
        self._comm = None
        self._native_half = None
        self.force_collectives = bool(force_collectives)
        self.compress = compress
        self.module = module
        self.optimizer = optimizer
        self.group = group
        self.max_norm = max_norm
        self.bucket_floats = bucket_floats
        self.sync_bn = bool(sync_bn)
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._reducer = None
        self._stream = None

    @property
    def stream(self):
        """The stream the step's kernels run on (HIP devices): a HIGH-priority stream of the driver's
        own.  HIP maps the streams of one priority level onto a few hardware queues, and streams on
        one queue run in submission order; torch's process group launches its RCCL kernels from a
        normal-priority pool stream, which as often as not shares its queue with torch's default
        stream — the bucket all-reduces (which wait for the side stream's weight gradients) and the
        main-stream kernels then serialise each other: 1.40 instead of 1.14 ms per step
        (profiles/r03_dp_overhead.md).  With compute on the high level, the weight gradients on the
        library's lowest-level side stream and the collectives on the normal level, the three never
        share a queue.  ``train_step`` hops onto this stream and back; a loop that runs entirely
        under ``with torch.cuda.stream(dp.stream):`` skips the two hops."""
        dev = self.module.engine.device
        if dev is None:                      # (the engine binds to a device at its first call)
            dev = next(self.module.parameters()).device
        if dev.type != "cuda":
            return None
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        if self._stream is None or self._stream.device != dev:
            # one per device and process: every driver object shares it
            if dev not in _COMPUTE_STREAMS:
                _COMPUTE_STREAMS[dev] = torch.cuda.Stream(device=dev, priority=-1)
            self._stream = _COMPUTE_STREAMS[dev]
        return self._stream

    def _all_reduce_sum(self, tensor):
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group)

    def native_comm(self):
        """The library-owned communicator of ``collectives="native"`` (created at the first step: a collective call).
        Rank 0 draws the RCCL unique id, the existing process group carries its 128 bytes to the other ranks."""
        if self._comm is None:
            from . import _native as N
            eng = self.module.engine
            dev = eng.device if eng.device is not None else next(self.module.parameters()).device
            if dev.type != "cuda":
                raise RuntimeError("collectives='native' needs a HIP device (RCCL); use the torch mode over gloo")
            if dev.index is None:
                dev = torch.device("cuda", torch.cuda.current_device())
            if dist.is_initialized() and self.world > 1:
                on_dev = dist.get_backend(self.group) == "nccl"
                buf = torch.zeros(N.UNIQUE_ID_BYTES, dtype=torch.uint8, device=dev if on_dev else "cpu")
                if self.rank == 0:
                    buf.copy_(torch.frombuffer(bytearray(N.rccl_unique_id()), dtype=torch.uint8))
                src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
                dist.broadcast(buf, src=src, group=self.group)
                uid = bytes(buf.cpu().numpy().tobytes())
            else:
                uid = N.rccl_unique_id()
            self._comm = N.Comm(dev, uid, self.world, self.rank)
        return self._comm

    def broadcast_parameters(self):
        eng = self.module.engine
        if self.world > 1:
            dist.broadcast(eng.params, src=0, group=self.group)
            dist.broadcast(eng.bn_running, src=0, group=self.group)

    @torch.no_grad()
    def train_step(self, x, target):
        self.module.engine.ensure(x.device)
        st = self.stream
        if st is None or torch.cuda.current_stream(x.device) == st:
            return self._train_step(x, target)
        cur = torch.cuda.current_stream(x.device)
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            out = self._train_step(x, target)
        cur.wait_stream(st)
        return out

    def _train_step_native(self, x, target):
        """The whole data-parallel step as one native call (blh_train_step_dp)."""
        import ctypes

        from . import _native as N
        from .engine import OUT_FEATURES
        eng, opt = self.module.engine, self.optimizer
        comm = self.native_comm()
        x = eng._check_input(x)
        batch = x.shape[0]
        if batch < 2:
            raise ValueError("Expected more than 1 value per channel when training")
        target = target.contiguous()
        if tuple(target.shape) != (batch, OUT_FEATURES) or target.dtype != torch.float32:
            raise RuntimeError("bad target: %s %s" % (tuple(target.shape), target.dtype))
        want = self.bucket_floats if self.bucket_floats else max(1 << 20, eng.layout.total // 4)
        if getattr(eng, "_bucket_floats_set", None) != want:
            eng.ctx.set_option(N.OPT_BUCKET_FLOATS, min(int(want), (1 << 31) - 1))
            eng._bucket_floats_set = want
        eng.row_offset = self.rank * _row_stride(batch)
        ws = eng.workspace(batch)
        drop = eng._drop_struct(batch)
        eng._tune_streams()
        opt._ensure_moments(eng)
        g = opt.param_groups[0]
        hyper = N.AdamHyper(float(g["lr"]), g["betas"][0], g["betas"][1], g["eps"],
                            0.0 if self.max_norm is None else float(self.max_norm), opt._t + 1, 0)
        pred = torch.empty(batch, OUT_FEATURES, dtype=torch.float32, device=x.device)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        errors = []
        use_sync = self.sync_bn and (self.world > 1 or self.force_collectives)
        scb = eng._sync_callback(ws, self._all_reduce_sum, errors) if use_sync else ctypes.cast(None, N.SyncFn)
        flags = N.DP_TAIL_ON_COMM_STREAM if self.native_tail == "comm" else 0
        if self.compress == "bf16":      # bf16 buckets: the library casts, exchanges and reads this image of the gradient arena
            if self._native_half is None or self._native_half.device != x.device or \
                    self._native_half.numel() != eng.layout.total:
                self._native_half = torch.zeros(eng.layout.total, dtype=torch.bfloat16, device=x.device)
                N.check(N.lib().blh_comm_set_bf16_buffer(comm.handle, N.ptr(self._native_half), eng.layout.total),
                        "blh_comm_set_bf16_buffer")
            flags |= N.DP_BF16_BUCKETS
        eng._check_shadow_versions()     # (bf16 storage: the step's Adam keeps the bf16 weight image, as the fused step's)
        N.check(N.lib().blh_train_step_dp(
            eng.ctx.handle, comm.handle, ctypes.byref(eng.layout.desc), eng._stream(), N.ptr(eng.params),
            N.ptr(eng.grads), N.ptr(opt._exp_avg), N.ptr(opt._exp_avg_sq), N.ptr(eng.bn_running), N.ptr(eng.bn_nbt),
            N.ptr(x), N.ptr(target), ctypes.byref(drop), eng._momentum(), ctypes.byref(hyper), None, N.ptr(ws),
            ws.numel(), N.ptr(pred), N.ptr(loss), N.ptr(opt._stats), batch, batch * self.world, scb, None, flags),
            "blh_train_step_dp")
        if errors:
            raise errors[0]
        opt._t += 1                         # (only a step that was enqueued counts)
        eng.generation += 1
        eng._saved_batch = None
        if eng.masks is None:
            eng.rng_step += 1
        opt._sync_step_state(eng)
        return pred, loss

    def _train_step(self, x, target):
        if self.collectives == "native":
            return self._train_step_native(x, target)
        eng = self.module.engine
        opt = self.optimizer
        if self._reducer is None or self._reducer.flat.data_ptr() != eng.grads.data_ptr():
            self._reducer = GradBucketReducer(eng.grads, self.group, self.bucket_floats,
                                              force_collectives=self.force_collectives,
                                              compress=self.compress)
        # the library merges the ranges it reports into the reducer's buckets: the hook is entered once per bucket,
        # not once per stage (blh_context_set_option(BLH_OPT_BUCKET_FLOATS))
        from . import _native as N
        if getattr(eng, "_bucket_floats_set", None) != self._reducer.bucket_floats:
            eng.ctx.set_option(N.OPT_BUCKET_FLOATS, min(self._reducer.bucket_floats, (1 << 31) - 1))
            eng._bucket_floats_set = self._reducer.bucket_floats
        batch = x.shape[0]
        eng.row_offset = self.rank * _row_stride(batch)
        sync = self._all_reduce_sum if (self.sync_bn and (self.world > 1 or self.force_collectives)) else None
        gb = batch * self.world
        # forward + MSE as the fused single-GPU step runs them (one enqueue; the loss gradient stays
        # in the workspace for backward)
        pred, loss = eng.forward_train_loss(x, target, sync=sync, global_batch=gb)
        self._reducer.begin()
        # the reported loss is the GLOBAL batch's: mean of the per-rank means (equal shards; the
        # reference logs the loss of the whole batch, train_bilinear.py:86-88).  Its 1-float
        # all-reduce goes out HERE, in front of backward (the forward finished the scalar), so it
        # travels beside the backward GEMMs and the wait for the first gradient bucket covers it
        # (one collective stream, in order).  Launched behind the last bucket and awaited after
        # Adam — the round-3 order — it held the NEXT step's first kernel back by a collective
        # plus a cross-queue hop: 23 us per step at configs[1] (profiles/r05_dp_overhead.md)
        self._reducer.reduce_scalars(loss)
        eng.backward(x, None, on_ready=self._reducer.on_ready, sync=sync, global_batch=gb)
        # bf16 buckets on a HIP device stay bf16: norm, clip and Adam read them directly — when the
        # reducer really exchanged (its own state decides: ``force_collectives`` without a process
        # group launches nothing and leaves no bf16 image to read)
        self._reducer.flush()
        exchanged = (self._reducer.world > 1 or self._reducer.force) and self._reducer._half is not None
        half_direct = self.compress == "bf16" and eng.grads.is_cuda and exchanged
        gscale = self._reducer.finish(cast_back=not half_direct)
        opt._ensure_moments(eng)
        g = opt.param_groups[0]
        opt._t += 1
        if half_direct:
            eng.clip_adam_bf16(self._reducer._half, gscale, opt._exp_avg, opt._exp_avg_sq, float(g["lr"]),
                               g["betas"], g["eps"], self.max_norm, opt._t, opt._stats)
        else:
            eng.clip_adam(opt._exp_avg, opt._exp_avg_sq, float(g["lr"]), g["betas"], g["eps"],
                          self.max_norm, opt._t, opt._stats)
        opt._sync_step_state(eng)
        return pred, loss


class CapturedDataParallelStep:
    """The data-parallel step as ONE hipGraph (BASELINE configs[4]: "overlapped all-reduce +
    hipGraph-captured train step"): forward, MSE, backward with the bucket all-reduces launched
    from the grad-ready hook (RCCL calls are captured like kernels: torch's process group joins the
    capture through events), clip + Adam.  Everything that changes per step lives in device memory
    (``blh_step_state``: Adam step count and bias corrections, learning rate, dropout step —
    advanced by the first node of the graph), the batch in static input buffers.

        step = CapturedDataParallelStep(dp, per_rank_batch)
        pred, loss = step(x_local, t_local)

    BatchNorm statistics are per rank (SyncBN's host callbacks are not captured)."""

    def __init__(self, dp, batch):
        import ctypes

        from . import _native as N
        if dp.sync_bn:
            raise RuntimeError("SyncBN is not capturable; use per-rank statistics")
        self.dp, self.batch = dp, int(batch)
        eng, opt = dp.module.engine, dp.optimizer
        dev = next(dp.module.parameters()).device
        eng.ensure(dev)
        opt._ensure_moments(eng)
        self.eng, self.opt = eng, opt
        self.x = torch.zeros(batch, 32, dtype=torch.float32, device=dev)
        self.t = torch.zeros(batch, 48, dtype=torch.float32, device=dev)
        self.state = torch.zeros(ctypes.sizeof(N.StepState), dtype=torch.uint8, device=dev)
        self._mirror = None
        self._write_state()
        eng.workspace(batch)
        eng.scratch()
        snap = [t.clone() for t in (eng.params, opt._exp_avg, opt._exp_avg_sq, eng.bn_running, eng.bn_nbt)]
        # collectives="native", bf16 storage: the captured form re-casts the parameter image in every replay (a
        # replay is invisible to the context's record of whose image is current; the eager native step keeps it)
        keep_image = None
        if dp.collectives == "native":
            if dp.compress is not None:
                raise RuntimeError("bf16 buckets are not capturable with the library-driven collectives "
                                   "(blh_train_step_dp: BLH_DP_BF16_BUCKETS needs the host-side hyper-parameters)")
            dp.native_comm()
            if eng.layout.desc.gemm_dtype == 4:
                keep_image = eng.ctx.get_option(N.OPT_PERSISTENT_SHADOW)
                eng.ctx.set_option(N.OPT_PERSISTENT_SHADOW, 0)
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):              # warm-up (kernel attributes, RCCL channels)
                self._enqueue()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            for dst, src in zip((eng.params, opt._exp_avg, opt._exp_avg_sq, eng.bn_running, eng.bn_nbt), snap):
                dst.copy_(src)
            self._write_state()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.pred, self.loss = self._enqueue()
        finally:
            if keep_image is not None:
                eng.ctx.set_option(N.OPT_PERSISTENT_SHADOW, keep_image)

    def _write_state(self):
        from . import _native as N
        g = self.opt.param_groups[0]
        rng = int(self.eng.rng_step)
        st = N.StepState(float(g["lr"]), g["betas"][0], g["betas"][1], g["eps"],
                         0.0 if self.dp.max_norm is None else float(self.dp.max_norm),
                         rng - 1 if rng > 0 else 2 ** 64 - 1, int(self.opt._t), 0, 0.0, 0.0)
        host = torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8)
        self.state.copy_(host)                      # synchronous: rare (lr changes, out-of-band steps)
        self._mirror = (float(g["lr"]), int(self.opt._t), rng)

    def _enqueue(self):
        import ctypes

        from . import _native as N
        dp, eng, opt = self.dp, self.eng, self.opt
        lib = N.lib()
        if dp.collectives == "native":
            return self._enqueue_native()
        if dp._reducer is None or dp._reducer.flat.data_ptr() != eng.grads.data_ptr():
            dp._reducer = GradBucketReducer(eng.grads, dp.group, dp.bucket_floats,
                                            force_collectives=dp.force_collectives, compress=dp.compress)
        N.check(lib.blh_step_state_advance(eng._stream(), N.ptr(self.state)), "blh_step_state_advance")
        N.check(lib.blh_context_set_step_state(eng.ctx.handle, N.ptr(self.state)), "blh_context_set_step_state")
        try:
            eng.row_offset = dp.rank * _row_stride(self.batch)
            saved_step = eng.rng_step
            eng.rng_step = 0                        # the device counter supplies the step
            pred, loss = eng.forward_train_loss(self.x, self.t)
            dp._reducer.begin()
            dp._reducer.reduce_scalars(loss)       # (in front of backward: see DataParallel._train_step)
            eng.backward(self.x, None, on_ready=dp._reducer.on_ready)
            dp._reducer.finish()
            eng.rng_step = saved_step
        finally:
            N.check(lib.blh_context_set_step_state(eng.ctx.handle, None), "blh_context_set_step_state")
        sc = eng.scratch()
        N.check(lib.blh_clip_adam_step_captured(
            eng._stream(), N.ptr(eng.params), N.ptr(eng.grads), N.ptr(opt._exp_avg), N.ptr(opt._exp_avg_sq),
            eng.layout.total, N.ptr(self.state), N.ptr(sc), sc.numel(), N.ptr(opt._stats)),
            "blh_clip_adam_step_captured")
        return pred, loss

    def _enqueue_native(self):
        """The step as ONE library call with the per-step scalars read from ``self.state`` (blh_train_step_dp with
        dev_state: it advances the state itself; the RCCL launches are captured like kernels)."""
        import ctypes

        from . import _native as N
        from .engine import OUT_FEATURES
        dp, eng, opt = self.dp, self.eng, self.opt
        want = dp.bucket_floats if dp.bucket_floats else max(1 << 20, eng.layout.total // 4)
        if getattr(eng, "_bucket_floats_set", None) != want:
            eng.ctx.set_option(N.OPT_BUCKET_FLOATS, min(int(want), (1 << 31) - 1))
            eng._bucket_floats_set = want
        ws = eng.workspace(self.batch)
        eng.row_offset = dp.rank * _row_stride(self.batch)
        saved_step = eng.rng_step
        eng.rng_step = 0                            # the device counter supplies the step
        try:
            drop = eng._drop_struct(self.batch)
        finally:
            eng.rng_step = saved_step
        eng._tune_streams()
        pred = torch.empty(self.batch, OUT_FEATURES, dtype=torch.float32, device=self.x.device)
        loss = torch.empty((), dtype=torch.float32, device=self.x.device)
        flags = N.DP_TAIL_ON_COMM_STREAM if dp.native_tail == "comm" else 0
        N.check(N.lib().blh_train_step_dp(
            eng.ctx.handle, dp.native_comm().handle, ctypes.byref(eng.layout.desc), eng._stream(), N.ptr(eng.params),
            N.ptr(eng.grads), N.ptr(opt._exp_avg), N.ptr(opt._exp_avg_sq), N.ptr(eng.bn_running), N.ptr(eng.bn_nbt),
            N.ptr(self.x), N.ptr(self.t), ctypes.byref(drop), eng._momentum(), None, N.ptr(self.state), N.ptr(ws),
            ws.numel(), N.ptr(pred), N.ptr(loss), N.ptr(opt._stats), self.batch, self.batch * dp.world,
            ctypes.cast(None, N.SyncFn), None, flags), "blh_train_step_dp")
        return pred, loss

    @torch.no_grad()
    def __call__(self, x, target):
        want = (float(self.opt.param_groups[0]["lr"]), int(self.opt._t), int(self.eng.rng_step))
        if want != self._mirror:
            self._write_state()
        if x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x, non_blocking=True)
        if target.data_ptr() != self.t.data_ptr():
            self.t.copy_(target, non_blocking=True)
        self.graph.replay()
        self.opt._t += 1
        self.opt._sync_step_state(self.eng)
        self.eng.rng_step += 1
        self.eng._saved_batch = None
        self.eng.generation += 1
        self.eng.invalidate_shadow()        # (the replay's Adam moved the parameters behind any bf16 image a fused step kept)
        self._mirror = (self._mirror[0], int(self.opt._t), int(self.eng.rng_step))
        return self.pred, self.loss
