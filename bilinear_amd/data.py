"""Input side of the lifter (/root/reference/H36M/data.py, train_bilinear.py:33-43,72-73).

The reference keeps the Human3.6M annotations on the host, z-scores one sample at a time in
``Dataset.__getitem__`` and pushes every mini-batch through a DataLoader (8 worker processes,
pinned memory) and a host-to-device copy.  At > 3 M poses/s that pipeline is the bottleneck, so
here the whole split lives on the device: ``DevicePoseDataset`` holds the flat, already z-scored
``[N,32]`` / ``[N,48]`` tensors (a few hundred MB for Human3.6M), shuffles with an on-device
permutation and hands out batches by one gather — no worker, no copy, no host synchronisation.

``DevicePoseDataset.from_pickles`` reads the reference's ``{task}_{protocol}.bin`` files when a
Human3.6M directory is available; ``synthetic_raw`` produces raw annotations of the same layout
(there is no dataset in this environment), so both enter through the same preprocessing."""
from __future__ import annotations

import os
import pickle

import numpy as np
import torch

ACTIONS = ["Directions", "Discussion", "Eating", "Greeting", "Phoning", "Photo", "Posing",
           "Purchases", "Sitting", "SittingDown", "Smoking", "Waiting", "WalkDog", "Walking",
           "WalkTogether"]


def decode_action(image_name):
    """'S1_Directions_1.54138969_000001.jpg' -> 'Directions' (H36M/util.py:13-22 followed by the
    sub-action merge of valid_bilinear.py:64)."""
    parts = image_name.split(".")[0].split("_")
    return parts[1]


class DevicePoseDataset:
    """One split (train or valid), resident on ``device`` and normalised with the TRAIN statistics.

        train = DevicePoseDataset(raw_train, device)                        # computes the statistics
        valid = DevicePoseDataset(raw_valid, device, stats_from=train)
        for x, t in train.epoch(epoch, batch_size, shuffle=True): ...
        for x, t, action_ids in valid.epoch(0, batch_size, with_actions=True): ...

    ``raw`` is the reference's pickle payload: ``{'part': [n,17,2], 'S': [n,17,3], 'image': names}``
    (lists or arrays; 'center' / 'scale' are image-only and ignored)."""

    def __init__(self, raw, device, stats_from=None, seed=0):
        self.device = torch.device(device)
        part = torch.as_tensor(np.asarray(raw["part"], dtype=np.float32)).to(self.device)
        S = torch.as_tensor(np.asarray(raw["S"], dtype=np.float32)).to(self.device)
        if part.dim() != 3 or part.shape[1:] != (17, 2) or S.shape[1:] != (17, 3) or len(part) != len(S):
            raise ValueError("expected part [n,17,2] and S [n,17,3], got %s and %s" % (
                tuple(part.shape), tuple(S.shape)))
        # H36M/data.py:41-43: 16 joints, the nose (index 9) is dropped
        keep = [j for j in range(17) if j != 9]
        x = part[:, keep, :].reshape(-1, 32)
        # H36M/data.py:46-54: root-centred, then the (all-zero) pelvis is dropped
        t = (S - S[:, 0:1, :])[:, 1:, :].reshape(-1, 48)
        # H36M/data.py:57-59: np.mean / np.std (population) per feature, in float32 like numpy
        self.mean_x, self.std_x = x.mean(0), x.std(0, unbiased=False)
        self.mean_t, self.std_t = t.mean(0), t.std(0, unbiased=False)
        src = stats_from if stats_from is not None else self
        # H36M/data.py:108-110: every split is z-scored with the statistics of the TRAIN split
        self.x = ((x - src.mean_x) / src.std_x).contiguous()
        self.t = ((t - src.mean_t) / src.std_t).contiguous()
        self.norm_mean, self.norm_stddev = src.mean_t, src.std_t     # de-normalisation of the metric
        names = raw.get("image")
        if names is not None:
            idx = {a: i for i, a in enumerate(ACTIONS)}
            self.action_names = list(ACTIONS)
            ids = []
            for n in names:
                a = decode_action(n)
                if a not in idx:
                    idx[a] = len(self.action_names)
                    self.action_names.append(a)
                ids.append(idx[a])
            self.action_ids = torch.tensor(ids, dtype=torch.int32, device=self.device)
        else:
            self.action_names, self.action_ids = list(ACTIONS), None
        self.seed = int(seed)

    def __len__(self):
        return self.x.shape[0]

    def num_batches(self, batch_size, drop_last=False):
        n = len(self)
        return n // batch_size if drop_last else (n + batch_size - 1) // batch_size

    def epoch(self, epoch, batch_size, shuffle=False, drop_last=False, with_actions=False):
        """Batches of one pass over the split (DataLoader(shuffle=..., drop_last=False) semantics,
        train_bilinear.py:33-43).  The permutation is drawn on the device from (seed, epoch)."""
        n = len(self)
        xs, ts, acts = self.x, self.t, self.action_ids
        if shuffle:
            # the whole epoch is gathered ONCE (two launches, 0.2 ms for Human3.6M's 1.56 M poses) and the batches are
            # views of it: at the reference's batch of 64 the step takes 0.15 ms, and a per-batch index_select pair
            # (two launches + 40 us of host time) would cost more than the step it feeds
            g = torch.Generator(device=self.device).manual_seed(self.seed * 1000003 + int(epoch))
            perm = torch.randperm(n, device=self.device, generator=g)
            xs, ts = xs.index_select(0, perm), ts.index_select(0, perm)
            if with_actions and acts is not None:
                acts = acts.index_select(0, perm)
        for b in range(self.num_batches(batch_size, drop_last)):
            lo, hi = b * batch_size, min(n, (b + 1) * batch_size)
            x, t = xs[lo:hi], ts[lo:hi]
            a = acts[lo:hi] if (with_actions and acts is not None) else None
            yield (x, t, a) if with_actions else (x, t)

    @staticmethod
    def from_pickles(data_dir, device, protocol="GT", seed=0):
        """(train, valid) from the reference's files ``{data_dir}/{task}_{protocol}.bin``
        (H36M/data.py:31-34)."""
        if protocol not in ("GT", "SH", "SH+FT"):      # H36M/data.py:22 asserts the same set
            raise ValueError("protocol must be 'GT', 'SH' or 'SH+FT', got %r" % (protocol,))
        raws = {}
        for task in ("train", "valid"):
            path = os.path.join(data_dir, "%s_%s.bin" % (task, protocol))
            with open(path, "rb") as f:
                raws[task] = pickle.load(f)
        train = DevicePoseDataset(raws["train"], device, seed=seed)
        valid = DevicePoseDataset(raws["valid"], device, stats_from=train, seed=seed)
        return train, valid


def synthetic_raw(n, seed=0):
    """Raw annotations with the layout of the reference's pickles (no Human3.6M here): a random
    skeleton in millimetres seen by a pinhole camera, 17 joints, names that decode to the 15
    actions.  The 2D joints are a (non-linear) function of the 3D ones, so the lifter can learn."""
    rng = np.random.RandomState(seed)
    root = np.concatenate([rng.uniform(-800, 800, (n, 1, 2)), rng.uniform(3500, 6000, (n, 1, 1))], axis=2)
    offs = rng.standard_normal((n, 17, 3)) * np.array([250.0, 350.0, 200.0])
    offs[:, 0, :] = 0.0
    S = (root + offs).astype(np.float32)
    f, c = 1145.0, 500.0
    part = (f * S[:, :, :2] / S[:, :, 2:3] + c + rng.standard_normal((n, 17, 2)) * 0.5).astype(np.float32)
    names = ["S%d_%s%s.%d_%06d.jpg" % (1 + i % 7, ACTIONS[i % len(ACTIONS)], "_1" if (i // 15) % 2 else "",
                                       54138969 + i % 4, i) for i in range(n)]
    return {"part": part, "S": S, "image": names,
            "center": [None] * n, "scale": [1.0] * n}


class SyntheticPoses:
    """Stream of random normalised batches (no dataset object): used by the quick script modes."""

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
