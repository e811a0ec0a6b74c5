"""Developer tool (not collected by pytest): the GLOBAL batches of BASELINE configs[3] / configs[4] on one GPU — 65536 and
131072 rows, far beyond what any test runs — through the size-independent property of tests/test_gpu_timed_path.py: a batch
made of N copies of a small batch (inputs, targets, dropout masks repeated) has the same BatchNorm statistics, the same
mean-reduced loss and the same parameter gradients as the small batch.  Index arithmetic, grid limits, workspace carving
and the split plans at these sizes are what is being exercised (the kernels' numerics are covered elsewhere).

    python tests/large_batch_check.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bilinear_amd  # noqa: E402

dev = torch.device("cuda", 0)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def grads_of(net, x, t, masks, fused):
    net.train()
    net.engine.set_dropout_masks(masks)
    if fused:
        opt = bilinear_amd.Adam(net.parameters(), lr=0.0, module=net)      # lr 0: the step leaves the parameters alone
        pred, loss = net.train_step(opt, x, t, max_norm=1e30)
    else:
        for p in net.parameters():
            p.grad = None
        pred = net(x)
        loss = torch.nn.functional.mse_loss(pred, t)
        loss.backward()
    torch.cuda.synchronize()
    eng = net.engine
    g = {name: v.clone() for (name, _, _, _), v in zip(eng._named_params(), eng.grad_views())}
    return pred.detach(), float(loss.item()), g


def main():
    bad = 0
    for dtype, nb, width, small, copies, tol in (("fp32", 2, 1024, 4096, 8, 2e-4), ("bf16s", 4, 1024, 8192, 8, 2e-2),
                                                 ("bf16s", 4, 1024, 16384, 4, 2e-2), ("bf16s", 8, 2048, 16384, 8, 2e-2)):
        t0 = time.time()
        torch.manual_seed(1)
        net, _, _, _ = bilinear_amd.load(dev, num_blocks=nb, width=width, gemm_dtype=dtype)
        g = torch.Generator(device=dev).manual_seed(2)
        x = torch.randn(small, 32, device=dev, generator=g)
        t = torch.randn(small, 48, device=dev, generator=g)
        nh = 1 + 2 * nb
        masks = [(torch.rand(small, width, device=dev, generator=g) < 0.5).to(torch.uint8) for _ in range(nh)]
        for fused in (True, False):
            p1, l1, g1 = grads_of(net, x, t, masks, fused)
            big_masks = [m.repeat(copies, 1) for m in masks]
            p2, l2, g2 = grads_of(net, x.repeat(copies, 1), t.repeat(copies, 1), big_masks, fused)
            del big_masks
            errs = {k: rel(g2[k], g1[k]) for k in g1 if not (k.endswith(".0.bias") and not k.startswith("decode"))}
            worst = max(errs.items(), key=lambda kv: kv[1])
            e_pred = rel(p2[:small], p1)
            ok = e_pred <= tol and abs(l2 - l1) <= tol * abs(l1) and worst[1] <= (20 * tol if dtype == "bf16s" else tol) and \
                all(torch.isfinite(v).all() for v in g2.values())
            bad += 0 if ok else 1
            print("%-5s %d x %4d  %6d rows = %d x %5d (%s): pred %.2e loss %.2e worst gradient %.2e (%s)%s  [%.0fs]" % (
                dtype, nb, width, small * copies, copies, small, "fused step" if fused else "drop-in", e_pred,
                abs(l2 - l1) / abs(l1), worst[1], worst[0], "" if ok else "  <-- FAILED", time.time() - t0), flush=True)
            del p2, g2
            torch.cuda.empty_cache()
        del net
        torch.cuda.empty_cache()
    print("%d failed" % bad, flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
