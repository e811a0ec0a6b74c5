#!/usr/bin/env python3
"""Counterpart of /root/reference/valid_bilinear.py: restore the newest checkpoint,
eval-mode forward (:31,52), de-normalise with the train-set mean/std (:53-54), MPJPE
= mean over joints of the Euclidean distance, reported per action and on average
(:56-83).  Synthetic data stand in for Human3.6M (see train_bilinear.py)."""
import argparse
import logging
import os

import torch

import bilinear_amd
from bilinear_amd.data import ACTIONS, DevicePoseDataset, SyntheticPoses, synthetic_raw
from bilinear_amd.metrics import MPJPE

COMMENT = "Bilinear GT"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--batch-size", type=int, default=64)
    ap.add_argument("--save-root", default="save")
    ap.add_argument("--data-dir", default=None, help="Human3.6M directory with train_GT.bin / valid_GT.bin")
    ap.add_argument("--protocol", choices=["GT", "SH", "SH+FT"], default="GT",
                    help="which 2D input the pickles hold (/root/reference/H36M/protocol.py:1-4): ground truth, "
                         "stacked-hourglass detections, fine-tuned detections; selects {task}_{protocol}.bin")
    ap.add_argument("--synthetic-poses", type=int, default=0,
                    help="no dataset: N synthetic raw annotations through the same device pipeline "
                         "(use the value train_bilinear.py was run with)")
    args = ap.parse_args()
    logging.basicConfig(level=logging.INFO)
    logger = logging.getLogger("valid_bilinear")
    if not torch.cuda.is_available():
        raise SystemExit("valid_bilinear.py needs a HIP device (MI355X)")
    device = torch.device("cuda")
    parameter_dir = os.path.join(args.save_root, COMMENT, "parameter")
    bilinear, optimizer, step, train_epoch = bilinear_amd.load(device=device, parameter_dir=parameter_dir)
    bilinear.eval()

    valid = None
    if args.data_dir:
        _, valid = DevicePoseDataset.from_pickles(args.data_dir, device, protocol=args.protocol)
    elif args.synthetic_poses:
        train = DevicePoseDataset(synthetic_raw(args.synthetic_poses, seed=0), device)
        valid = DevicePoseDataset(synthetic_raw(max(1, args.synthetic_poses // 4), seed=1), device, stats_from=train)
    if valid is not None:
        # the whole split, its train-set statistics and the action ids are device tensors
        metric = MPJPE(valid.action_names, valid.norm_mean, valid.norm_stddev, device)
        with torch.set_grad_enabled(False):
            for in_image_space, in_camera_space, ids in valid.epoch(0, args.batch_size, with_actions=True):
                metric.update(bilinear(in_image_space), in_camera_space, ids)
        per_action, average = metric.result()
        for key, value in per_action.items():
            logger.info("%s: %f", key, value)
        logger.info("avg: %f", average)
        return

    data = SyntheticPoses(args.steps, args.batch_size, device, seed=999)
    metric = MPJPE(ACTIONS, data.mean, data.stddev, device)   # per-action sums stay on the device
    with torch.set_grad_enabled(False):
        for in_image_space, in_camera_space, mean, stddev, action in data.epoch(0, with_stats=True):
            prediction = bilinear(in_image_space)
            ids = torch.tensor([ACTIONS.index(a.split("_")[0]) for a in action], dtype=torch.int32)
            metric.update(prediction, in_camera_space, ids)
    per_action, average = metric.result()
    for key, value in per_action.items():
        logger.info("%s: %f", key, value)
    logger.info("avg: %f", average)


if __name__ == "__main__":
    main()
