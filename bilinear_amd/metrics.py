"""Validation metric of the reference on the device
(/root/reference/valid_bilinear.py:53-83): MPJPE per action and on average."""
from __future__ import annotations

import ctypes

import torch

from . import _native as N

NUM_JOINT = 16


class MPJPE:
    """Accumulates per-action distance sums over batches without host synchronisation.

        m = MPJPE(action_names, mean, stddev, device)
        m.update(prediction, ground_truth, action_ids)     # normalised [B,48] tensors, int32 ids
        per_action, average = m.result()                   # mm
    """

    def __init__(self, action_names, mean, stddev, device):
        self.names = list(action_names)
        self.mean = mean.to(device=device, dtype=torch.float32).contiguous()
        self.stddev = stddev.to(device=device, dtype=torch.float32).contiguous()
        self.sum = torch.zeros(len(self.names), dtype=torch.float64, device=device)
        self.count = torch.zeros(len(self.names), dtype=torch.int64, device=device)

    def update(self, prediction, ground_truth, action_ids):
        if prediction.device.type != "cuda":
            raise RuntimeError("bilinear_amd.metrics runs only on a HIP device")
        batch = prediction.shape[0]
        dist = torch.empty(batch, dtype=torch.float32, device=prediction.device)
        ids = action_ids.to(device=prediction.device, dtype=torch.int32).contiguous()
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        N.check(N.lib().blh_mpjpe(st, N.ptr(prediction.contiguous()), N.ptr(ground_truth.contiguous()),
                                  N.ptr(self.mean), N.ptr(self.stddev), batch, NUM_JOINT,
                                  N.ptr(dist), N.ptr(ids), len(self.names), N.ptr(self.sum),
                                  N.ptr(self.count)), "blh_mpjpe")
        return dist

    def result(self):
        s, c = self.sum.cpu(), self.count.cpu()
        per_action = {n: float(s[i] / (c[i] * NUM_JOINT)) for i, n in enumerate(self.names) if c[i] > 0}
        avg = float(s.sum() / (c.sum() * NUM_JOINT)) if int(c.sum()) else float("nan")
        return per_action, avg
