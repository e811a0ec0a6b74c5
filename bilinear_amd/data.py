"""Synthetic stand-in for the reference's Human3.6M loader (/root/reference/H36M/data.py):
same output contract for the lifter — per-feature z-scored fp32 vectors, 32 wide
(2D joints) and 48 wide (root-relative 3D joints), H36M/data.py:108-110 — but generated
on the device, so no DataLoader / host-to-device copy sits in front of the step."""
from __future__ import annotations

import torch

ACTIONS = ["Directions", "Discussion", "Eating", "Greeting", "Phoning", "Photo", "Posing",
           "Purchases", "Sitting", "SittingDown", "Smoking", "Waiting", "WalkDog", "Walking",
           "WalkTogether"]


class SyntheticPoses:
    def __init__(self, steps_per_epoch, batch_size, device, seed=0):
        self.steps = steps_per_epoch
        self.batch = batch_size
        self.device = device
        self.seed = seed
        # a fixed random linear map 2D -> 3D plus noise gives the network something to learn
        g = torch.Generator(device="cpu").manual_seed(seed + 17)
        self.lift = (torch.randn(32, 48, generator=g) / 32 ** 0.5).to(device)
        self.mean = (torch.randn(48, generator=g) * 100).to(device)
        self.stddev = (torch.rand(48, generator=g) * 200 + 50).to(device)

    def epoch(self, epoch, with_stats=False):
        g = torch.Generator(device=self.device).manual_seed(self.seed * 100003 + epoch)
        for i in range(self.steps):
            x = torch.randn(self.batch, 32, device=self.device, generator=g)
            t = x @ self.lift + 0.1 * torch.randn(self.batch, 48, device=self.device, generator=g)
            if with_stats:
                action = [ACTIONS[(i + j) % len(ACTIONS)] for j in range(self.batch)]
                yield x, t, self.mean, self.stddev, action
            else:
                yield x, t
