#!/usr/bin/env python3
"""Developer tool: checksum of the parameters after N fused steps from a fixed seed, twice in one process:
run-to-run determinism in and across processes.  usage: step_checksum.py <config 1..4> [steps]"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bilinear_amd  # noqa: E402

CFG = {1: (2, 1024, 4096, "fp32"), 2: (4, 1024, 16384, "bf16s"), 3: (4, 1024, 8192, "bf16s"), 4: (8, 2048, 16384, "bf16s")}


def run(cfg, steps, dev):
    nb, w, b, dt = CFG[cfg]
    torch.manual_seed(1)
    net, opt, _, _ = bilinear_amd.load(dev, num_blocks=nb, width=w, gemm_dtype=dt)
    net.train()
    g = torch.Generator(device=dev).manual_seed(1000)
    x = torch.randn(b, 32, device=dev, generator=g)
    t = torch.randn(b, 48, device=dev, generator=g)
    sums = []
    for s in range(steps):
        _, loss = net.train_step(opt, x, t, max_norm=1.0)
        if s in (0, 1, 9, steps - 1):
            torch.cuda.synchronize()
            sums.append((s, hashlib.md5(net.engine.params.cpu().numpy().tobytes()).hexdigest()[:10], float(loss.item())))
    return sums


def main():
    cfg = int(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    dev = torch.device("cuda", 0)
    for rep in range(2):
        print("config %d rep %d:" % (cfg, rep), run(cfg, steps, dev), flush=True)


if __name__ == "__main__":
    main()
