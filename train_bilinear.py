#!/usr/bin/env python3
"""Counterpart of the reference's training script (/root/reference/train_bilinear.py) on
the MI355X implementation.  Same step order — lr-decay check on the pre-increment
step (:66-70), zero_grad, forward, MSELoss, backward, clip_grad_norm_(1), Adam.step
(:75-83) — same checkpoint dict and path (:92-104), 10 epochs per invocation (:56).

Human3.6M, dotmap, tensorboardX and vectormath are not available here, so the data
come from a synthetic loader with the real loader's output contract (z-scored fp32
[B,32] / [B,48], H36M/data.py:108-110) and the config values of util/config.py:13-25
are plain constants below.

    python train_bilinear.py [--fast] [--epochs N] [--steps-per-epoch N] [--log-every N]

--fast uses BilinearUnit.train_step (one native enqueue per step) instead of the
reference's five separate calls; the numerics are the same.

Every step's loss is reported, as the reference does (:86-88: add_scalar + set_postfix) — but
through a device ring read back once per --log-every steps (bilinear_amd.LossRing), not with a
``loss.item()`` synchronisation per step: the lines "step N loss L" go to
{save-root}/{comment}/loss.log (tensorboardX is not available here).
"""
import argparse
import logging
import os

import torch
import torch.nn as nn

import bilinear_amd
from bilinear_amd import config          # util/config.py:13-25 (one place: the tests import the same lambdas)
from bilinear_amd.data import DevicePoseDataset, SyntheticPoses, synthetic_raw

COMMENT = config.bilinear.comment
BATCH_SIZE = config.bilinear.batch_size


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fast", action="store_true")
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--steps-per-epoch", type=int, default=200)
    ap.add_argument("--batch-size", type=int, default=BATCH_SIZE)
    ap.add_argument("--save-root", default="save")
    ap.add_argument("--data-dir", default=None,
                    help="Human3.6M directory holding the reference's train_GT.bin / valid_GT.bin; "
                         "the split is preprocessed once and kept on the device")
    ap.add_argument("--protocol", choices=["GT", "SH", "SH+FT"], default="GT",
                    help="which 2D input the pickles hold (/root/reference/H36M/protocol.py:1-4): ground truth, "
                         "stacked-hourglass detections, fine-tuned detections; selects {task}_{protocol}.bin")
    ap.add_argument("--synthetic-poses", type=int, default=0,
                    help="no dataset: N synthetic raw annotations through the same device pipeline")
    ap.add_argument("--log-every", type=int, default=100,
                    help="steps between two read-backs of the per-step losses (0: no per-step log)")
    args = ap.parse_args()

    logging.basicConfig(level=logging.INFO, format="[%(levelname)s|%(filename)s:%(lineno)s] %(asctime)s > %(message)s")
    logger = logging.getLogger("train_bilinear")
    if not torch.cuda.is_available():
        raise SystemExit("train_bilinear.py needs a HIP device (MI355X); there is no CPU path")
    device = torch.device("cuda")
    log_dir = os.path.join(args.save_root, COMMENT)
    parameter_dir = os.path.join(log_dir, "parameter")

    # H36M.Dataset + DataLoader(shuffle=True) (train_bilinear.py:33-43) as one device-resident split
    dataset = None
    if args.data_dir:
        dataset, _ = DevicePoseDataset.from_pickles(args.data_dir, device, protocol=args.protocol)
    elif args.synthetic_poses:
        dataset = DevicePoseDataset(synthetic_raw(args.synthetic_poses, seed=0), device)
    data = SyntheticPoses(args.steps_per_epoch, args.batch_size, device)
    bilinear, optimizer, step, train_epoch = bilinear_amd.load(
        device=device, parameter_dir=parameter_dir if os.path.exists(parameter_dir) else None)
    criterion = nn.MSELoss()
    bilinear.train()
    # The five-call step is host-bound at the reference's batch of 64 (GPU: 0.15 ms of kernels per step).  torch's
    # backward normally hands the graph to the autograd engine's device thread and waits for it; one operator deep,
    # that hand-off is the largest single host cost of the step and the noisiest (0.30-0.56 ms per step measured on
    # different boxes against a steady 0.28 on the calling thread, bench.py batch_64.five_call_drop_in).
    bilinear_amd.configure_for_small_batches()
    os.makedirs(log_dir, exist_ok=True)
    loss_file = open(os.path.join(log_dir, "loss.log"), "a")
    ring = bilinear_amd.LossRing(device, every=args.log_every,
                                 sink=lambda s, v: loss_file.write("step %d loss %.9g\n" % (s, v))) \
        if args.log_every > 0 else None
    logger.info("Training resumes at epoch %d (step %d)", train_epoch + 1, step)

    for epoch in range(train_epoch + 1, train_epoch + args.epochs + 1):
        loss = None
        batches = (dataset.epoch(epoch, args.batch_size, shuffle=True) if dataset is not None
                   else data.epoch(epoch))
        for in_image_space, in_camera_space in batches:
            if config.bilinear.lr_decay.activate and config.bilinear.lr_decay.condition(step):
                lr = config.bilinear.lr_decay.function(step)
                logger.info("Learning rate decay to %s (step: %d)", lr, step)
                for param_group in optimizer.param_groups:
                    param_group["lr"] = lr
            if args.fast:
                _, loss = bilinear.train_step(optimizer, in_image_space, in_camera_space, max_norm=1.0,
                                              loss_out=ring.slot() if ring is not None else None)
            else:
                optimizer.zero_grad()
                prediction = bilinear(in_image_space)
                loss = criterion(prediction, in_camera_space)
                loss.backward()
                bilinear_amd.clip_grad_norm_(bilinear.parameters(), max_norm=1, module=bilinear)
                optimizer.step()
                if ring is not None:
                    ring.push(loss)
            if ring is not None:
                ring.advance(step)
            step = step + 1
        if ring is not None:
            ring.flush()
            loss_file.flush()
        os.makedirs(parameter_dir, exist_ok=True)
        torch.save({"epoch": epoch, "step": step, "state": bilinear.state_dict(),
                    "optimizer": optimizer.state_dict()},
                   os.path.join(parameter_dir, "%d.save" % epoch))
        logger.info("Epoch %d saved (loss: %f)", epoch, float(loss.item()))


if __name__ == "__main__":
    main()
