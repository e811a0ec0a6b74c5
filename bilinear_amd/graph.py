"""hipGraph-captured training step.

The whole step body of /root/reference/train_bilinear.py:75-83 is a fixed DAG of ~50
kernels with no host synchronisation and no allocation, so it is captured
once into a HIP graph and replayed with one launch per step.  Everything that changes
from step to step lives in device memory: the Adam step count, the learning rate and the
dropout step are fields of a ``blh_step_state`` that the first node of the graph advances
(include/bilinear_hip.h), the batch is copied into static input buffers.  This is what
makes the reference's own batch size (64, util/config.py:15) launch-latency free.

PyTorch provides the capture machinery (``torch.cuda.CUDAGraph`` is hipGraph on ROCm);
every node of the graph is a kernel of libbilinear_hip.so.
"""
from __future__ import annotations

import ctypes

import torch

from . import _native as N


class CapturedTrainStep:
    """``step = CapturedTrainStep(net, opt, batch); pred, loss = step(x, target)``.

    ``pred`` / ``loss`` are static device tensors overwritten by every replay.  The
    learning rate is read from ``opt.param_groups[0]['lr']`` before each replay (the
    lr-decay hook keeps working); ``opt`` state (exp_avg, exp_avg_sq, step) stays in sync.
    """

    def __init__(self, module, optimizer, batch, max_norm=1.0, device=None, two_stream=False,
                 persistent_shadow=None):
        """``two_stream=False`` (default): the graph is captured in the single-stream kernel order.
        Measured on MI355X (profiles/r02_step_timeline.md): hipGraph replay spreads the forked
        weight-gradient branch over up to four hardware queues and every cross-queue edge costs
        10-18 us, so the two-stream DAG replays SLOWER (1.171 ms at B=4096) than its own
        single-stream capture (1.141 ms) — eager enqueue, which keeps both streams on two queues
        with CP-pipelined dispatch, is the fast form there (1.10 ms).  At the reference's batch
        size (64) the step is bound by the ~5 us minimum duration of each of its ~55 kernels, and
        replay equals eager (0.34 ms)."""
        eng = module.engine
        dev = torch.device(device) if device is not None else next(module.parameters()).device
        eng.ensure(dev)
        optimizer._ensure_moments(eng)
        # gemm_dtype "bf16s": Adam writes the bf16 weight image, the graph holds no arena re-cast
        # (Engine.set_persistent_shadow; the image is refreshed here whenever the host-side mirror
        # shows that something else moved the training state)
        # ``persistent_shadow``: None = the engine's current setting (on by default for bf16 storage); True / False
        # set it for the engine (a graph replays what was captured: the option is fixed at capture time)
        if eng.layout.desc.gemm_dtype != 4:
            self.persistent_shadow = False
        else:
            if persistent_shadow is not None:
                eng.set_persistent_shadow(bool(persistent_shadow))
            self.persistent_shadow = bool(eng.ctx.get_option(N.OPT_PERSISTENT_SHADOW))
        self._param_versions = None
        if eng.masks is not None:
            raise RuntimeError("explicit dropout masks cannot be captured; use the Philox stream")
        self.module, self.opt, self.eng, self.batch = module, optimizer, eng, int(batch)
        self.max_norm = 0.0 if max_norm is None else float(max_norm)
        self.x = torch.zeros(batch, 32, dtype=torch.float32, device=dev)
        self.t = torch.zeros(batch, 48, dtype=torch.float32, device=dev)
        self.pred = torch.empty(batch, 48, dtype=torch.float32, device=dev)
        self.loss = torch.empty((), dtype=torch.float32, device=dev)
        self.ws = eng.workspace(batch)
        self.state = torch.zeros(ctypes.sizeof(N.StepState), dtype=torch.uint8, device=dev)
        # two pinned staging buffers used alternately, each guarded by the event of its last
        # H2D copy: rewriting the state on consecutive steps never races an in-flight copy
        self._host_state = [torch.zeros(ctypes.sizeof(N.StepState), dtype=torch.uint8).pin_memory()
                            for _ in range(2)]
        self._host_event = [None, None]
        self._host_slot = 0
        self._mirror = None            # (lr, opt._t, eng.rng_step) the device state was written for
        self._write_state()
        self._drop = N.Dropout(None, eng.seed, 0, eng.row_offset, 0, 0)
        # Warm-up on a side stream (sets kernel attributes, touches every buffer) on a snapshot
        # of the training state, which is restored afterwards: building the graph must not
        # move the model.  Capture itself executes nothing.
        was_two = eng.ctx.get_option(N.OPT_TWO_STREAM)
        eng.ctx.set_option(N.OPT_TWO_STREAM, 1 if two_stream else 0)
        snap = [t.clone() for t in (eng.params, optimizer._exp_avg, optimizer._exp_avg_sq,
                                    eng.bn_running, eng.bn_nbt)]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self._enqueue()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        for dst, src in zip((eng.params, optimizer._exp_avg, optimizer._exp_avg_sq,
                             eng.bn_running, eng.bn_nbt), snap):
            dst.copy_(src)
        self._write_state()
        self._refresh_shadow()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._enqueue()
        eng.ctx.set_option(N.OPT_TWO_STREAM, was_two)

    # -- device step state ------------------------------------------------------
    def _write_state(self):
        g = self.opt.param_groups[0]
        st = N.StepState(float(g["lr"]), g["betas"][0], g["betas"][1], g["eps"], self.max_norm,
                         int(self.eng.rng_step) - 1 if self.eng.rng_step > 0 else 2 ** 64 - 1,
                         int(self.opt._t), 0, 0.0, 0.0)
        # rng_step holds (next dropout step - 1): blh_step_state_advance adds 1 before use
        raw = bytes(st)
        slot = self._host_slot
        self._host_slot ^= 1
        if self._host_event[slot] is not None:
            self._host_event[slot].synchronize()
        self._host_state[slot].copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
        self.state.copy_(self._host_state[slot], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._host_event[slot] = ev
        self._mirror = (float(g["lr"]), int(self.opt._t), int(self.eng.rng_step))

    def _refresh_shadow(self):
        self._shadow_epoch = self.eng.shadow_epoch
        if self.persistent_shadow:
            eng = self.eng
            N.check(N.lib().blh_refresh_param_shadow(
                eng.ctx.handle, ctypes.byref(eng.layout.desc), eng._stream(), N.ptr(eng.params),
                N.ptr(self.ws), self.ws.numel(), self.batch), "blh_refresh_param_shadow")

    def _enqueue(self):
        eng, opt = self.eng, self.opt
        N.check(N.lib().blh_train_step_captured(
            eng.ctx.handle, ctypes.byref(eng.layout.desc), eng._stream(), N.ptr(eng.params), N.ptr(eng.grads),
            N.ptr(opt._exp_avg), N.ptr(opt._exp_avg_sq), N.ptr(eng.bn_running), N.ptr(eng.bn_nbt),
            N.ptr(self.x), N.ptr(self.t), ctypes.byref(self._drop), eng._momentum(),
            N.ptr(self.state), N.ptr(self.ws), self.ws.numel(), N.ptr(self.pred),
            N.ptr(self.loss), N.ptr(opt._stats), self.batch), "blh_train_step_captured")

    def _after_step(self):
        self.opt._t += 1
        self.opt._sync_step_state(self.eng)
        self.eng.rng_step += 1
        self.eng._saved_batch = None
        self.eng.generation += 1
        # the graph's first node advanced the device counters exactly like this
        self._mirror = (self._mirror[0], int(self.opt._t), int(self.eng.rng_step))

    @torch.no_grad()
    def __call__(self, x, target):
        if x.shape[0] != self.batch:
            raise RuntimeError("captured for batch %d, got %d" % (self.batch, x.shape[0]))
        if not self.eng.is_packed(self.x.device) or self.eng.workspace(self.batch) is not self.ws:
            raise RuntimeError("the module's device arenas changed after capture; re-capture")
        # the device-resident (lr, Adam step, dropout step) must be what the host expects: an
        # eager step, optimizer.step() or another captured step in between moves the host side
        want = (float(self.opt.param_groups[0]["lr"]), int(self.opt._t), int(self.eng.rng_step))
        if self.persistent_shadow:
            # a Parameter written in place since the last replay (its version counter moved): the image is stale
            versions = sum(p._version for _, p, _, _ in self.eng._named_params())
            if versions != self._param_versions:
                self._param_versions = versions
                self.eng.invalidate_shadow()
        if want != self._mirror:
            self._write_state()
            self._refresh_shadow()          # an out-of-band step moved the parameters too
        elif self._shadow_epoch != self.eng.shadow_epoch:
            self._refresh_shadow()          # a checkpoint load / re-initialisation wrote the parameters
        # (callers that fill ``self.x`` / ``self.t`` in place and pass them back pay no copy)
        if x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x, non_blocking=True)
        if target.data_ptr() != self.t.data_ptr():
            self.t.copy_(target, non_blocking=True)
        self.graph.replay()
        self._after_step()
        return self.pred, self.loss
