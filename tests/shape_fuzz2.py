"""Developer tool (uses the oracle; not collected by pytest): the surfaces tests/shape_fuzz.py does not reach, at odd shapes.

  eval     — eval-mode forward (running statistics, no dropout) against the fp64 oracle, batches 1 ... 4100
  graph    — CapturedTrainStep (hipGraph) against the eager fused step, three steps: bit-identical state
  heavy    — one stand-alone heavy_linear stage (blh_heavy_forward / _backward) against plain PyTorch on the CPU, odd
             feature counts (multiples of 4) and batches

    python tests/shape_fuzz2.py [eval] [graph] [heavy]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bilinear_amd  # noqa: E402
from oracle import numpy_oracle as O  # noqa: E402

dev = torch.device("cuda", 0)
bad = 0
t0 = time.time()


def report(what, ok, detail):
    global bad
    bad += 0 if ok else 1
    print("%-60s %s%s  [%.0fs]" % (what, detail, "" if ok else "  <-- FAILED", time.time() - t0), flush=True)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def build(dtype, nb, width, seed):
    st = O.init_state(seed, nb, width)
    # (running statistics away from their initial 0 / 1, so that eval mode has something to normalise with)
    rng = np.random.default_rng(seed)
    for k in st:
        if k.endswith("running_mean"):
            st[k] = rng.normal(0, 0.3, st[k].shape).astype(np.float32)
        if k.endswith("running_var"):
            st[k] = rng.uniform(0.5, 2.0, st[k].shape).astype(np.float32)
    net = bilinear_amd.BilinearUnit(nb, width, gemm_dtype=dtype)
    sd = net.state_dict()
    net.load_state_dict({k: torch.from_numpy(np.array(st[k])).reshape(sd[k].shape) for k in sd})
    return st, net.to(dev)


def fuzz_eval():
    for dtype, tol in (("fp32", 2e-5), ("bf16s", 2e-2)):
        for nb, width in ((1, 256), (2, 1024), (3, 512)):
            st, net = build(dtype, nb, width, 5)
            net.eval()
            for b in (1, 2, 3, 5, 31, 64, 65, 383, 384, 385, 1000, 1025, 2049, 4100):
                x, _ = O.synthetic_batch(50 + b, b)
                with torch.no_grad():
                    got = net(torch.from_numpy(x).to(dev)).cpu().numpy()
                if dtype == "bf16s":
                    O.set_gemm_rounding("bf16s")
                try:
                    ref, _ = O.forward({k: v.copy() for k, v in st.items()}, x, None, training=False, dtype=np.float64)
                finally:
                    O.set_gemm_rounding(None)
                e = rel(got, ref)
                report("eval %-5s %d x %4d B = %4d" % (dtype, nb, width, b), e <= tol and np.isfinite(got).all(), "rel %.2e" % e)


def fuzz_graph():
    for dtype in ("fp32", "bf16s"):
        for nb, width, b in ((2, 512, 385), (2, 512, 1025), (2, 1024, 2049), (1, 256, 4100), (2, 1024, 64), (2, 512, 37)):
            x = torch.randn(b, 32, device=dev, generator=torch.Generator(device=dev).manual_seed(b))
            t = torch.randn(b, 48, device=dev, generator=torch.Generator(device=dev).manual_seed(b + 1))
            outs = []
            for captured in (False, True):
                torch.manual_seed(3)
                net, opt, _, _ = bilinear_amd.load(dev, num_blocks=nb, width=width, gemm_dtype=dtype)
                net.train()
                net.engine.seed = 11
                step = bilinear_amd.CapturedTrainStep(net, opt, b, max_norm=1.0) if captured else None
                losses = []
                for i in range(3):
                    if i == 2:
                        opt.param_groups[0]["lr"] = 5e-4
                    _, loss = step(x, t) if captured else net.train_step(opt, x, t, max_norm=1.0)
                    losses.append(float(loss.item()))
                torch.cuda.synchronize()
                outs.append((net.engine.params.clone(), net.engine.bn_running.clone(), opt._exp_avg_sq.clone(), losses))
            same = all(torch.equal(a, c) for a, c in zip(outs[0][:3], outs[1][:3])) and outs[0][3] == outs[1][3]
            report("graph %-5s %d x %4d B = %4d: captured == eager" % (dtype, nb, width, b), same, "losses %s" % outs[1][3])


def fuzz_heavy():
    for fin, fout, b in ((4, 4, 2), (12, 36, 5), (36, 100, 33), (64, 192, 300), (100, 260, 129), (32, 1024, 385), (1024, 48, 1025),
                         (516, 516, 2049), (1028, 260, 513)):
        torch.manual_seed(fin + fout)
        stage = bilinear_amd.heavy_linear(fin, fout).to(dev).train()
        ref = torch.nn.Sequential(torch.nn.Linear(fin, fout), torch.nn.BatchNorm1d(fout), torch.nn.ReLU())
        ref[0].load_state_dict({k: v.cpu() for k, v in stage[0].state_dict().items()})
        with torch.no_grad():
            stage[1].weight.uniform_(0.5, 1.5)
            stage[1].bias.normal_(0, 0.2)
        ref[1].load_state_dict({k: v.cpu() for k, v in stage[1].state_dict().items()})
        ref = ref.double().train()
        x = torch.randn(b, fin)
        xg = x.to(dev).requires_grad_(True)
        out = stage(xg)
        xr = x.double().requires_grad_(True)
        y = ref(xr)
        keep = (out.detach().cpu() != 0) | (y.detach() <= 0)
        yr = y * keep.double() * 2.0
        g = torch.randn(b, fout)
        out.backward(g.to(dev))
        yr.backward(g.double())
        errs = dict(out=rel(out.detach().cpu().numpy(), yr.detach().numpy()),
                    dx=rel(xg.grad.cpu().numpy(), xr.grad.numpy()),
                    dW=rel(stage[0].weight.grad.cpu().numpy(), ref[0].weight.grad.numpy()),
                    dgamma=rel(stage[1].weight.grad.cpu().numpy(), ref[1].weight.grad.numpy()),
                    dbeta=rel(stage[1].bias.grad.cpu().numpy(), ref[1].bias.grad.numpy()),
                    rvar=rel(stage[1].running_var.cpu().numpy(), ref[1].running_var.numpy()))
        tol = 1e-3 if b < 8 else 2e-4
        report("heavy_linear %4d -> %4d B = %4d" % (fin, fout, b), max(errs.values()) <= tol,
               " ".join("%s %.1e" % kv for kv in errs.items()))


def fuzz_mpjpe():
    """blh_mpjpe (valid_bilinear.py:53-83 on the device) at odd batch sizes, missing actions, one-row updates."""
    from bilinear_amd.metrics import MPJPE
    rng = np.random.RandomState(3)
    names = ["a%d" % i for i in range(15)]
    for B in (1, 2, 15, 16, 17, 63, 64, 65, 255, 257, 777, 1000, 4097, 10000):
        pred = rng.standard_normal((B, 48)).astype(np.float32)
        tgt = rng.standard_normal((B, 48)).astype(np.float32)
        mean = (rng.standard_normal(48) * 100).astype(np.float32)
        std = (50 + 200 * rng.random_sample(48)).astype(np.float32)
        ids = rng.randint(0, rng.randint(1, 16), size=B).astype(np.int32)        # (some actions may not occur)
        m = MPJPE(names, torch.from_numpy(mean), torch.from_numpy(std), dev)
        cuts = sorted(set([0, B] + [int(c) for c in rng.randint(0, B + 1, size=3)]))
        got = []
        for a, b in zip(cuts, cuts[1:]):
            if b > a:
                got.append(m.update(torch.from_numpy(pred[a:b]).to(dev), torch.from_numpy(tgt[a:b]).to(dev),
                                    torch.from_numpy(ids[a:b])).cpu().numpy())
        got = np.concatenate(got)
        ref = O.mpjpe_sum(pred.astype(np.float64), tgt.astype(np.float64), mean.astype(np.float64), std.astype(np.float64))
        per_action, avg = m.result()
        ok = np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max() and abs(avg - ref.sum() / (B * 16)) <= 1e-5 * avg
        for i, n in enumerate(names):
            sel = ids == i
            if sel.sum():
                ok = ok and abs(per_action[n] - ref[sel].sum() / (sel.sum() * 16)) <= 1e-5 * per_action[n]
        report("mpjpe B = %5d in %d updates" % (B, len(cuts) - 1), bool(ok), "avg %.4f" % avg)


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if a in ("eval", "graph", "heavy", "mpjpe")] or ["eval", "graph", "heavy", "mpjpe"]
    for w in which:
        {"eval": fuzz_eval, "graph": fuzz_graph, "heavy": fuzz_heavy, "mpjpe": fuzz_mpjpe}[w]()
    print("%d failed" % bad, flush=True)
    sys.exit(1 if bad else 0)
