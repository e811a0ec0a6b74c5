"""Developer tool (uses the oracle, so it lives under tests/; not collected by pytest): two fused training steps on the
device against the fp64 NumPy oracle over a grid of ODD shapes — batches around every dispatch boundary of the library
(small-batch kernels <= 384 rows, split-K <= 1024, 64-row tiles <= 2048, ragged row tiles, batches that are not
multiples of 8 / 32 / 128), widths 256 / 512 / 1024, 1-3 blocks, fp32 and bf16 storage, explicit masks.

    python tests/shape_fuzz.py [quick] [big] [wide] [split] [fused] [dropin] [dp] [dpsync]   -> one line per case, exit code 1 if any case is
                                                                    outside its tolerance

Masks are gate-safe (golden_util.safe_masks: elements whose ReLU gate sits within rounding of zero are dropped, so no
gradient depends on which way a correct implementation rounds).  Tolerances: fp32 — predictions 1e-4, loss 1e-5,
gradients 1e-3 relative L2; bf16 storage — 1.5e-2 / 1e-3 / 6e-2 against the SAME-ROUNDING oracle.  Pre-BatchNorm biases
(mathematically zero gradients) are skipped; batches of 2 rows are left out (BatchNorm over two samples is +-1 whatever
the input: the gradients are differences of nearly equal numbers in any arithmetic).

Round 6: this grid found that every ragged batch above 384 rows ran its weight-gradient slabs past their buffer in bf16
storage (api_layout.h: slab_floats_h) — a zero loss and a wrong decode-bias gradient from the fused step, a memory fault
at 4100 rows.  tests/test_gpu_shape_fuzz.py runs a subset of it in the suite."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bilinear_amd  # noqa: E402
from golden_util import is_prebn_bias, safe_masks  # noqa: E402
from oracle import numpy_oracle as O  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def case(dev, dtype, nb, width, batch, seed, mode="fused"):
    """mode: "fused" (blh_train_step), "dropin" (the reference's five calls on the drop-in surface), "dp" (the
    data-parallel driver at one rank: blh_forward_train_loss + blh_backward under the bucket hook + clip + Adam) or
    "dpsync" (the same with SyncBN, every exchange issued over a one-rank gloo group)."""
    st = O.init_state(seed, nb, width)
    net = bilinear_amd.BilinearUnit(nb, width, gemm_dtype=dtype)
    sd = net.state_dict()
    net.load_state_dict({k: torch.from_numpy(np.array(st[k])).reshape(sd[k].shape) for k in sd})
    net = net.to(dev).train()
    opt = bilinear_amd.Adam(net.parameters(), lr=1e-3, module=net)
    x, t = O.synthetic_batch(seed + 1, batch)
    x = (x * 1.7 + 0.3).astype(np.float32)                  # (not standardised: the encode stage's shifted moments)
    keys = O.param_keys(nb)
    ost = {k: v.copy() for k, v in st.items()}
    oopt = O.adam_init(ost, keys)
    worst = dict(pred=0.0, loss=0.0, grad=0.0, gname="")
    xt, tt = torch.from_numpy(x).to(dev), torch.from_numpy(t).to(dev)
    dp = None
    if mode in ("dp", "dpsync"):
        from bilinear_amd.dp import DataParallel
        kw = {}
        if mode == "dpsync":     # SyncBN at one rank with every exchange issued (gloo): the materialised kernels, statistics
            import torch.distributed as dist      # through sum / sum of squares
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29577")
                dist.init_process_group("gloo", rank=0, world_size=1)
            kw = dict(sync_bn=True, force_collectives=True)
        dp = DataParallel(net, opt, bucket_floats=max(1 << 16, net.engine.layout.total // 4), **kw)
    for step in range(2):
        masks = safe_masks(ost, x, O.random_masks(seed + 10 + step, batch, nb, width),
                           "bf16s" if dtype == "bf16s" else None, thr=2e-2 if dtype == "bf16s" else 1e-4)
        net.engine.set_dropout_masks(masks)
        # re-synchronise the device with the oracle's state (two correct runs drift apart by lr * sign flips)
        sd = net.state_dict()
        net.load_state_dict({k: torch.from_numpy(np.array(ost[k])).reshape(sd[k].shape) for k in sd})
        if mode == "fused":
            pred, loss = net.train_step(opt, xt, tt, max_norm=1.0)
        elif mode in ("dp", "dpsync"):
            pred, loss = dp.train_step(xt, tt)
        else:
            opt.zero_grad()
            pred = net(xt)
            loss = torch.nn.functional.mse_loss(pred, tt)
            loss.backward()
            bilinear_amd.clip_grad_norm_(net.parameters(), max_norm=1, module=net)
            opt.step()
            pred = pred.detach()
        torch.cuda.synchronize()
        if dtype == "bf16s":
            O.set_gemm_rounding("bf16s")
        try:
            r = O.train_step(ost, oopt, x, t, masks, 1e-3, dtype=np.float64)
        finally:
            O.set_gemm_rounding(None)
        worst["pred"] = max(worst["pred"], rel(pred.cpu().numpy(), r["pred"]))
        worst["loss"] = max(worst["loss"], abs(loss.item() - r["loss"]) / r["loss"])
        eng = net.engine                      # (the fused step leaves the CLIPPED gradients in the arena, as .grad would hold)
        g = {name: v.detach().cpu().numpy() for (name, _, _, _), v in zip(eng._named_params(), eng.grad_views())}
        for k in keys:
            if is_prebn_bias(k):
                continue
            e = rel(g[k], r["grads"][k])
            if e > worst["grad"]:
                worst["grad"], worst["gname"] = e, k
        assert np.isfinite(pred.cpu().numpy()).all() and all(np.isfinite(v).all() for v in g.values())
    return worst


TOLERANCES = {"fp32": (1e-4, 1e-5, 1e-3), "bf16s": (1.5e-2, 1e-3, 6e-2),        # (prediction, loss, gradient)
              "bf16x3": (1e-4, 1e-5, 1e-3), "fp16x2": (1e-4, 1e-5, 1e-3)}


def main():
    args = sys.argv[1:]
    quick = "quick" in args
    big = "big" in args
    wide = "wide" in args        # widths off the beaten path (materialised encode / decode forms, other tile counts), 0-2 blocks
    split = "split" in args      # the fp32-accurate split modes on the 16-bit matrix cores
    modes = [m for m in ("fused", "dropin", "dp", "dpsync") if m in args] or ["fused"]
    dev = torch.device("cuda", 0)
    batches = [3, 37, 64, 129, 384, 385, 386, 388, 511, 1000, 1024, 1025, 1536, 2047, 2049, 2176, 3000, 4100, 4104]
    if quick:
        batches = [3, 37, 385, 1025, 2049, 4100]
    grid = []
    for mode in modes:
        for dtype in (("bf16x3", "fp16x2") if split else ("fp32", "bf16s")):
            if wide:
                for nb, width in ((0, 256), (0, 1024), (1, 128), (1, 384), (2, 640), (1, 768), (1, 1536), (2, 2048)):
                    for b in (37, 385, 1025, 2049, 4100):
                        grid.append((mode, dtype, nb, width, b))
                continue
            if split:
                for nb, width in ((2, 512), (2, 1024)):
                    for b in (37, 385, 1025, 2049, 4100):
                        grid.append((mode, dtype, nb, width, b))
                continue
            if big:      # the big-tile kernels with ragged row tiles (bf16 storage: 256 x 256 / 128 x 256 tiles; W = 2048)
                shapes = [(4, 1024, 8200), (4, 1024, 16392), (2, 2048, 4100), (2, 2048, 8200)] if dtype == "bf16s" \
                    else [(2, 1024, 8200), (1, 2048, 4100)]
                grid += [(mode, dtype, nb, w, b) for nb, w, b in shapes]
                continue
            for nb, width in ((1, 256), (2, 512), (2, 1024), (3, 1024)):
                if (quick or mode != "fused") and (nb, width) not in ((2, 512), (2, 1024)):
                    continue
                for b in batches:
                    if b < 8 and nb >= 3:      # (BatchNorm over three samples, seven stages deep: rounding noise in any arithmetic —
                        continue               #  fp32 2.3e-3, bf16 storage > 1 against the fp64 oracle)
                    grid.append((mode, dtype, nb, width, b))
    tol = TOLERANCES
    bad = 0
    t0 = time.time()
    for i, (mode, dtype, nb, width, b) in enumerate(grid):
        try:
            w = case(dev, dtype, nb, width, b, seed=100 + i, mode=mode)
            tp, tl, tg = tol[dtype]
            ok = w["pred"] <= tp and w["loss"] <= tl and w["grad"] <= tg
            note = ""
        except Exception as exc:       # noqa: BLE001  (a refused shape is a finding too)
            ok, w, note = False, dict(pred=-1, loss=-1, grad=-1, gname=""), " EXC %s: %s" % (type(exc).__name__, str(exc)[:100])
        bad += 0 if ok else 1
        print("%-6s %-5s %d x %4d  B = %5d: pred %.2e loss %.2e grad %.2e (%s)%s%s  [%.0fs]" % (
            mode, dtype, nb, width, b, w["pred"], w["loss"], w["grad"], w["gname"], "" if ok else "  <-- OUT OF TOLERANCE", note,
            time.time() - t0), flush=True)
    print("%d cases, %d out of tolerance" % (len(grid), bad), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
