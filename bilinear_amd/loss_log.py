"""Per-step loss reporting without a per-step synchronisation.

The reference reports every step's loss (``writer.add_scalar('BI/loss', loss, step)`` and
``progress.set_postfix(loss=float(loss.item()))``, /root/reference/train_bilinear.py:86-88): each ``loss.item()`` is a
device synchronisation, 0.15 ms of kernels per step wait for the host every step.  Here the step's loss lands in a
slot of a device ring — the fused step writes it there itself (``Engine.train_step(loss_out=...)``: no extra launch;
the five-call loop copies the scalar, one tiny device-to-device copy) — and the ring is read back ONCE every ``every``
steps: the same per-step values reach the log, ``every`` steps late at most.
"""
from __future__ import annotations

import torch


class LossRing:
    """``slot()`` -> 0-dim device tensor for the next step's loss; ``advance(step)`` after the step; every
    ``every`` steps (and at ``flush()``) ``sink(step, loss)`` is called for each buffered step in order."""

    def __init__(self, device, every=100, sink=None):
        self.every = int(every)
        if self.every < 1:
            raise ValueError("every must be >= 1")
        self.buf = torch.zeros(self.every, dtype=torch.float32, device=device)
        self.steps = []
        self.sink = sink
        self.last = None            # the newest loss read back (float), for the epoch line

    def slot(self):
        return self.buf[len(self.steps)]

    def push(self, loss):
        """Five-call loop: copy an existing scalar tensor into the next slot (asynchronous)."""
        self.buf[len(self.steps)].copy_(loss.detach(), non_blocking=True)

    def advance(self, step):
        self.steps.append(int(step))
        if len(self.steps) == self.every:
            self.flush()

    def flush(self):
        if not self.steps:
            return
        values = self.buf[:len(self.steps)].cpu().tolist()       # the one synchronisation per `every` steps
        if self.sink is not None:
            for s, v in zip(self.steps, values):
                self.sink(s, v)
        self.last = values[-1]
        self.steps = []
