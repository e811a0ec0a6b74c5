"""Developer tool (not collected by pytest): the drop-in surface under UNUSUAL user operations.  The host layer caches what
the five-call step asks for seven times per step (parameter lists, arena pointers, gradient views, module-tree links —
round 6's host trims); a stale cache would be silent.  A network goes through a random sequence of operations a PyTorch
user may perform between steps — grads set to None, zero_grad either way, a trip to the CPU and back, deepcopy,
state_dict round trips, re-initialisation, a Parameter replaced, torch.save / load of the optimizer — with drop-in steps
in between.  Then its state is copied into a FRESH network (no history, fresh caches) and both run the same two drop-in
steps with the same explicit dropout masks: parameters, statistics and Adam moments must come out bit-identical.

    python tests/dropin_fuzz.py [sequences] [ops]"""
import copy
import io
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bilinear_amd  # noqa: E402

dev = torch.device("cuda", 0)


def dropin_step(net, opt, x, t, masks=None):
    net.train()
    if masks is not None:
        net.engine.set_dropout_masks(masks)
    opt.zero_grad()
    loss = torch.nn.functional.mse_loss(net(x), t)
    loss.backward()
    bilinear_amd.clip_grad_norm_(net.parameters(), max_norm=1, module=net)
    opt.step()
    if masks is not None:
        net.engine.set_dropout_masks(None)
    return float(loss.item())


def run(nseq=10, nops=20):
    bad = 0
    for seq in range(nseq):
        rnd = random.Random(int(os.environ.get("FUZZ_SEED", "500")) + seq)
        dtype = rnd.choice(["fp32", "bf16s"])
        nb, width = rnd.choice([(1, 256), (2, 512), (2, 1024)])
        torch.manual_seed(seq)
        net, opt, _, _ = bilinear_amd.load(dev, num_blocks=nb, width=width, gemm_dtype=dtype)
        g = torch.Generator(device=dev).manual_seed(seq)
        log = []
        try:
            for _ in range(nops):
                b = rnd.choice([8, 64, 100, 385, 1024, 2049])
                x = torch.randn(b, 32, device=dev, generator=g)
                t = torch.randn(b, 48, device=dev, generator=g)
                op = rnd.choice(["step", "step", "step", "fused", "grad_none", "zero_false", "zero_none", "cpu_trip", "deepcopy",
                                 "reload", "reinit", "replace_param", "opt_roundtrip", "eval", "half_grad_none"])
                log.append((op, b))
                if op == "step":
                    dropin_step(net, opt, x, t)
                elif op == "fused":
                    net.train()
                    net.train_step(opt, x, t, max_norm=1.0)
                elif op == "grad_none":
                    for p in net.parameters():
                        p.grad = None
                elif op == "half_grad_none":
                    for i, p in enumerate(net.parameters()):
                        if i % 2:
                            p.grad = None
                elif op == "zero_false":
                    opt.zero_grad(set_to_none=False)
                elif op == "zero_none":
                    net.zero_grad(set_to_none=True)
                elif op == "cpu_trip":
                    net.to("cpu")
                    net.to(dev)
                elif op == "deepcopy":
                    sd, osd = copy.deepcopy(net.state_dict()), copy.deepcopy(opt.state_dict())
                    net2 = copy.deepcopy(net)
                    net = net2
                    opt = bilinear_amd.Adam(net.parameters(), lr=opt.param_groups[0]["lr"], module=net)
                    opt.load_state_dict(osd)
                    net.load_state_dict(sd)
                elif op == "reload":
                    net.load_state_dict({k: v.clone() for k, v in net.state_dict().items()})
                elif op == "reinit":
                    with torch.no_grad():
                        torch.nn.init.normal_(net.decode.weight, 0, 0.02)
                elif op == "replace_param":
                    net.decode.bias = torch.nn.Parameter(torch.zeros_like(net.decode.bias))
                    opt = bilinear_amd.Adam(net.parameters(), lr=opt.param_groups[0]["lr"], module=net)
                elif op == "opt_roundtrip":
                    buf = io.BytesIO()
                    torch.save(opt.state_dict(), buf)
                    buf.seek(0)
                    opt.load_state_dict(torch.load(buf, map_location=dev))
                elif op == "eval":
                    net.eval()
                    with torch.no_grad():
                        net(x)
                    net.train()
            # the fresh copy
            fresh, fopt, _, _ = bilinear_amd.load(dev, num_blocks=nb, width=width, gemm_dtype=dtype)
            fresh.load_state_dict({k: v.clone() for k, v in net.state_dict().items()})
            fopt.load_state_dict(copy.deepcopy(opt.state_dict()))
            b = 1000
            x = torch.randn(b, 32, device=dev, generator=g)
            t = torch.randn(b, 48, device=dev, generator=g)
            rng = np.random.default_rng(seq)
            nh = 1 + 2 * nb
            res = []
            for step in range(2):
                masks = [rng.integers(0, 2, (b, width)).astype(np.uint8) for _ in range(nh)]
                res.append((dropin_step(net, opt, x, t, masks), dropin_step(fresh, fopt, x, t, masks)))
            torch.cuda.synchronize()
            same = all(a == c for a, c in res) and all(torch.equal(p.detach(), q.detach()) for p, q in zip(net.parameters(), fresh.parameters()))
            same = same and all(torch.equal(v, fresh.state_dict()[k]) for k, v in net.state_dict().items())
            note = "" if same else " losses %s" % res
        except Exception as exc:      # noqa: BLE001
            same, note = False, " EXC %s: %s" % (type(exc).__name__, str(exc)[:300])
        bad += 0 if same else 1
        print("sequence %d (%s %d x %d): %s%s%s" % (seq, dtype, nb, width, "== a fresh copy" if same else "DIFFERS from a fresh copy", note,
                                                    "" if same else "  ops: %s" % log), flush=True)
    print("%d sequences, %d differ" % (nseq, bad), flush=True)
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else 20) else 0)
