#!/usr/bin/env python3
"""Developer tool: is the bf16-storage backward deterministic run to run?  Same state, same batch, N
backward passes: every gradient tensor must be bit-identical each time.  usage: k9_determinism.py [nb] [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bilinear_amd  # noqa: E402


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    width = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net, opt, _, _ = bilinear_amd.load(dev, num_blocks=nb, width=width, gemm_dtype="bf16s")
    net.train()
    x = torch.randn(batch, 32, device=dev)
    t = torch.randn(batch, 48, device=dev)
    eng = net.engine
    eng.ensure(dev)
    ref = None
    bad = 0
    for it in range(30):
        eng.rng_step = 0
        pred, loss = eng.forward_train_loss(x, t)
        eng.backward(x, None)
        torch.cuda.synchronize()
        g = eng.grads.clone()
        if ref is None:
            ref = g
            continue
        if not torch.equal(g, ref):
            bad += 1
            for name, off, shape in eng.layout.entries:
                n = 1
                for s in shape:
                    n *= s
                if not torch.equal(g[off:off + n], ref[off:off + n]):
                    d = (g[off:off + n] - ref[off:off + n]).abs().max().item()
                    print("iteration %d: %s differs (max abs %.3e)" % (it, name, d))
    print("nb=%d batch=%d width=%d: %d of 29 repeats differ from the first" % (nb, batch, width, bad))


if __name__ == "__main__":
    main()
