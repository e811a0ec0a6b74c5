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

The returned loss is the global batch's (a 1-float all-reduce behind the last bucket; every
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


class GradBucketReducer:
    """Bucketed, overlapped all-reduce (mean) over a flat gradient tensor.

    ``on_ready(offset, count)`` is called in backward order with contiguous ranges;
    ranges are merged until a bucket holds at least ``bucket_floats`` elements."""

    def __init__(self, flat_grads, group=None, bucket_floats=1 << 20, force_collectives=False):
        self.flat = flat_grads
        self.group = group
        self.bucket_floats = int(bucket_floats)
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

    def finish(self):
        if self._pending is not None:
            self._launch(*self._pending)
            self._pending = None
        for work, view in self._works:
            work.wait()
            if view is not None:
                view.div_(self.world)
        self._works = []


class DataParallel:
    """Data-parallel driver of BilinearUnit's fast path.

        dp = DataParallel(net, opt)            # after dist.init_process_group(...)
        pred, loss = dp.train_step(x_local, t_local)

    ``x_local`` is this rank's row slice of the global batch.  Dropout uses the
    global row index (rank * local_batch), so the Philox mask of a row does not
    depend on the number of GPUs."""

    def __init__(self, module, optimizer, group=None, bucket_floats=1 << 20, max_norm=1.0,
                 sync_bn=False, force_collectives=False):
        self.force_collectives = bool(force_collectives)
        self.module = module
        self.optimizer = optimizer
        self.group = group
        self.max_norm = max_norm
        self.bucket_floats = bucket_floats
        self.sync_bn = bool(sync_bn)
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._reducer = None

    def _all_reduce_sum(self, tensor):
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group)

    def broadcast_parameters(self):
        eng = self.module.engine
        if self.world > 1:
            dist.broadcast(eng.params, src=0, group=self.group)
            dist.broadcast(eng.bn_running, src=0, group=self.group)

    @torch.no_grad()
    def train_step(self, x, target):
        eng = self.module.engine
        opt = self.optimizer
        eng.ensure(x.device)
        if self._reducer is None or self._reducer.flat.data_ptr() != eng.grads.data_ptr():
            self._reducer = GradBucketReducer(eng.grads, self.group, self.bucket_floats,
                                              force_collectives=self.force_collectives)
        batch = x.shape[0]
        eng.row_offset = self.rank * batch
        sync = self._all_reduce_sum if (self.sync_bn and (self.world > 1 or self.force_collectives)) else None
        gb = batch * self.world
        pred = eng.forward_train(x, sync=sync, global_batch=gb)
        loss, dpred = eng.mse_loss_grad(pred, target)
        self._reducer.begin()
        eng.backward(x, dpred, on_ready=self._reducer.on_ready, sync=sync, global_batch=gb)
        # the reported loss is the GLOBAL batch's: mean of the per-rank means (equal shards), the
        # 1-float all-reduce rides behind the last gradient bucket (the reference
        # logs the loss of the whole batch, train_bilinear.py:86-88)
        self._reducer.reduce_scalars(loss)
        self._reducer.finish()
        opt._ensure_moments(eng)
        g = opt.param_groups[0]
        opt._t += 1
        eng.clip_adam(opt._exp_avg, opt._exp_avg_sq, float(g["lr"]), g["betas"], g["eps"],
                      self.max_norm, opt._t, opt._stats)
        opt._sync_step_state(eng)
        return pred, loss
