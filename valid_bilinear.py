#!/usr/bin/env python3
"""Counterpart of /root/reference/valid_bilinear.py: restore the newest checkpoint,
eval-mode forward (:31,52), de-normalise with the train-set mean/std (:53-54), MPJPE
= mean over joints of the Euclidean distance, reported per action and on average
(:56-83).  Synthetic data stand in for Human3.6M (see train_bilinear.py)."""
import argparse
import logging
import os

import numpy as np
import torch

import bilinear_amd
from bilinear_amd.data import SyntheticPoses

COMMENT = "Bilinear GT"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--batch-size", type=int, default=64)
    ap.add_argument("--save-root", default="save")
    args = ap.parse_args()
    logging.basicConfig(level=logging.INFO)
    logger = logging.getLogger("valid_bilinear")
    if not torch.cuda.is_available():
        raise SystemExit("valid_bilinear.py needs a HIP device (MI355X)")
    device = torch.device("cuda")
    parameter_dir = os.path.join(args.save_root, COMMENT, "parameter")
    bilinear, optimizer, step, train_epoch = bilinear_amd.load(device=device, parameter_dir=parameter_dir)
    bilinear.eval()

    data = SyntheticPoses(args.steps, args.batch_size, device, seed=999)
    total_dist, total = {}, {}
    with torch.set_grad_enabled(False):
        for in_image_space, in_camera_space, mean, stddev, action in data.epoch(0, with_stats=True):
            prediction = bilinear(in_image_space)
            prediction = (stddev * prediction + mean).view(-1, 16, 3)
            ground_truth = (stddev * in_camera_space + mean).view(-1, 16, 3)
            dist = torch.sum(torch.sqrt(torch.sum((prediction - ground_truth) ** 2, dim=2)), dim=1)
            dist = dist.double().cpu().numpy()
            for name in sorted(set(action)):
                sel = np.array([a == name for a in action])
                total_dist[name] = total_dist.get(name, 0.0) + float(dist[sel].sum())
                total[name] = total.get(name, 0) + int(sel.sum())
    dist_sum, cnt = 0.0, 0
    for key, value in total_dist.items():
        logger.info("%s: %f", key, value / (total[key] * 16))
        dist_sum += value
        cnt += total[key] * 16
    logger.info("avg: %f", dist_sum / cnt)


if __name__ == "__main__":
    main()
