"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

Plain-PyTorch (CPU) restatement of the reference's lifter network and training
step, generalised to ``(num_blocks, width)``.  Two uses only:

* a second checker: it is validated against the golden vectors captured from
  the reference (``tests/test_oracle_golden.py``), so that a CPU number timed
  with it is a number for the reference's own op sequence;
* the ``cpu_baseline`` leg of ``bench.py`` (kind "port"): the reference cannot
  travel to the GPU box, this file can.

It restates (not copies) ``/root/reference/model/bilinear.py:7-41`` — the op
sequence Linear -> BatchNorm1d -> ReLU -> Dropout(0.5), residual add around
pairs of such layers, Linear decode — with the same submodule names so that
``state_dict`` keys coincide (``encode.{0,1}.*``, ``bilinear.{b}.{l}.{0,1}.*``,
``decode.*``), and the step body of ``train_bilinear.py:75-83``.
Parity status: PINNED by the golden vectors.
"""
from __future__ import annotations

import time

import torch
from torch import nn

IN_FEATURES, OUT_FEATURES = 32, 48


def _stage(fan_in, fan_out):
    # model/bilinear.py:7-13
    return nn.Sequential(nn.Linear(fan_in, fan_out), nn.BatchNorm1d(fan_out),
                         nn.ReLU(), nn.Dropout(0.5))


class LifterPort(nn.Module):
    def __init__(self, num_blocks=2, width=1024):
        super().__init__()
        self.encode = _stage(IN_FEATURES, width)
        self.bilinear = nn.ModuleList(
            nn.Sequential(_stage(width, width), _stage(width, width))
            for _ in range(num_blocks))
        self.decode = nn.Linear(width, OUT_FEATURES)

    def forward(self, joints2d):
        h = self.encode(joints2d)
        for pair in self.bilinear:
            h = pair(h) + h                      # model/bilinear.py:35-38
        return self.decode(h)


def load_numpy_state(module, state):
    """Copy a dict of NumPy arrays keyed like the reference ``state_dict``."""
    sd = module.state_dict()
    assert list(sd.keys()) == list(state.keys()), "state_dict key mismatch"
    module.load_state_dict({k: torch.from_numpy(state[k].copy()).reshape(sd[k].shape)
                            for k in sd})


class MaskInjector:
    """Replace every Dropout by multiplication with a given keep-mask (x2), so a
    CPU run can replay the masks of a golden vector (PyTorch's Bernoulli stream
    is not reproducible elsewhere)."""

    def __init__(self, module):
        self.drops = [m for m in module.modules() if isinstance(m, nn.Dropout)]
        self.masks = None
        self.handles = [d.register_forward_hook(self._hook(i)) for i, d in enumerate(self.drops)]

    def _hook(self, i):
        def fn(mod, inp, out):
            if self.masks is None or not mod.training:
                return out
            return inp[0] * self.masks[i].to(inp[0].dtype) * 2.0
        return fn

    def remove(self):
        for h in self.handles:
            h.remove()


def train_step(module, optimizer, x, t):
    """train_bilinear.py:75-83."""
    optimizer.zero_grad()
    pred = module(x)
    loss = nn.functional.mse_loss(pred, t)
    loss.backward()
    total_norm = nn.utils.clip_grad_norm_(module.parameters(), max_norm=1)
    optimizer.step()
    return pred, loss, total_norm


def time_cpu_steps(num_blocks, width, batch, steps, warmup, threads=None, fwd_bwd_only=False):
    """Poses/s of the port on the host cores (used by bench.py cpu_baseline)."""
    if threads:
        torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)
    net = LifterPort(num_blocks, width)
    for m in net.modules():
        if isinstance(m, nn.Linear):
            nn.init.kaiming_normal_(m.weight)      # model/bilinear.py:86-90
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    x = torch.randn(batch, IN_FEATURES, generator=g)
    t = torch.randn(batch, OUT_FEATURES, generator=g)

    def one():
        if fwd_bwd_only:
            opt.zero_grad()
            nn.functional.mse_loss(net(x), t).backward()
        else:
            train_step(net, opt, x, t)

    for _ in range(warmup):
        one()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    dt = time.perf_counter() - t0
    return dict(poses_per_s=batch * steps / dt, ms_per_step=1e3 * dt / steps,
                threads=torch.get_num_threads(), steps=steps, batch=batch)
