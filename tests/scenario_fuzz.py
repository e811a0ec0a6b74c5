"""Developer tool (not collected by pytest): random SEQUENCES of operations on one bf16-storage network — fused steps at
changing batch sizes, the five-call drop-in step, data-parallel steps, eval forwards, lr changes, state_dict round trips —
run on two twins that differ ONLY in whether Adam keeps the bf16 weight image across steps (Engine.set_persistent_shadow).
The image is an optimisation that must never change a bit: after every operation the twins' parameters, Adam moments and
BatchNorm statistics are compared with torch.equal.  (Round 6: this is the kind of test that would have caught the
decode-weight image behind the batch-sized buffers — tests/test_gpu_bf16s.py pins that one sequence; this tool draws many.)

    python tests/scenario_fuzz.py [sequences] [ops per sequence] [shadow | streams | native | graph]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bilinear_amd  # noqa: E402
from bilinear_amd.dp import DataParallel  # noqa: E402

dev = torch.device("cuda", 0)
BATCHES = [37, 64, 100, 385, 1024, 1236, 2048, 2176, 4096, 4100]
COMPRESS = [None]          # ("native" twins: drawn per sequence — bf16 gradient buckets or fp32)


def make(first, nb, width, twin="shadow", dtype="bf16s"):
    """twin "shadow": the twins differ in whether Adam keeps the bf16 weight image; twin "streams": in whether the weight
    gradients run on the side stream (every schedule is bit-identical by construction: a race would show as a difference)."""
    torch.manual_seed(0)
    net, opt, _, _ = bilinear_amd.load(dev, num_blocks=nb, width=width, gemm_dtype=dtype)
    net.train()
    net.engine.seed = 9
    net.engine.ensure(dev)
    if twin == "shadow":
        net.engine.set_persistent_shadow(first)
    elif twin == "streams":
        net.engine.set_two_stream(first)
    if twin == "native":     # the data-parallel driver's collectives: torch's process group against the library's own RCCL calls
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29579")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        return net, opt, DataParallel(net, opt, force_collectives=True, collectives="torch" if first else "native",
                                      compress=COMPRESS[0])
    return net, opt, DataParallel(net, opt)


def apply(op, net, opt, dp, data, cap=None):
    kind, arg = op
    if kind == "cap":        # a fused step at the sequence's captured batch: the hipGraph replay on the first twin, eager on the second
        x, t = data[arg]
        net.train()
        if cap is not None:
            return float(cap(x, t)[1].item())
        return float(net.train_step(opt, x, t, max_norm=1.0)[1].item())
    if kind in ("fused", "dropin", "dp"):
        x, t = data[arg]
        net.train()
        if kind == "fused":
            return float(net.train_step(opt, x, t, max_norm=1.0)[1].item())
        if kind == "dp":
            return float(dp.train_step(x, t)[1].item())
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(net(x), t)
        loss.backward()
        bilinear_amd.clip_grad_norm_(net.parameters(), max_norm=1, module=net)
        opt.step()
        return float(loss.item())
    if kind == "eval":
        net.eval()
        with torch.no_grad():
            out = net(data[arg][0])
        net.train()
        return float(out.float().abs().mean().item())
    if kind == "lr":
        opt.param_groups[0]["lr"] = arg
        return arg
    if kind == "reload":
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        net.load_state_dict(sd)
        return 0.0
    if kind == "scale":      # an in-place write to a Parameter between steps (seen through its version counter)
        with torch.no_grad():
            dict(net.named_parameters())["decode.weight"].mul_(arg)
        return arg
    raise ValueError(kind)


def run(nseq=6, nops=25, twin="shadow"):
    """-> number of sequences whose twins diverged (0 = every operation left them bit-identical)"""
    g = torch.Generator(device=dev).manual_seed(3)
    data = {b: (torch.randn(b, 32, device=dev, generator=g), torch.randn(b, 48, device=dev, generator=g)) for b in BATCHES}
    bad = 0
    for seq in range(nseq):
        rnd = random.Random(int(os.environ.get("FUZZ_SEED", "100")) + seq)
        nb, width = rnd.choice([(1, 256), (2, 512), (2, 1024), (4, 1024)])
        ops = []
        for _ in range(nops):
            r = rnd.random()
            if r < 0.45:
                ops.append(("fused", rnd.choice(BATCHES)))
            elif r < 0.6:
                ops.append(("dropin", rnd.choice(BATCHES)))
            elif r < 0.72:
                ops.append(("dp", rnd.choice(BATCHES)))
            elif r < 0.84:
                ops.append(("eval", rnd.choice(BATCHES)))
            elif r < 0.9:
                ops.append(("lr", rnd.choice([1e-3, 5e-4, 2e-3])))
            elif r < 0.95:
                ops.append(("reload", None))
            else:
                ops.append(("scale", rnd.choice([0.99, 1.01])))
        dtype = "bf16s" if twin == "shadow" else rnd.choice(["fp32", "bf16s"])
        COMPRESS[0] = rnd.choice([None, "bf16"]) if twin == "native" else None
        if twin == "native":     # (mostly data-parallel steps)
            ops = [("dp", o[1]) if o[0] in ("fused", "dropin") and rnd.random() < 0.7 else o for o in ops]
        twins = [make(True, nb, width, twin, dtype), make(False, nb, width, twin, dtype)]
        caps = [None, None]
        if twin == "graph":      # twin 0 replays a captured step (its device-resident lr / step counts / bf16 image must follow
            b0 = rnd.choice([1024, 2048, 4096])        # every out-of-band change), twin 1 runs everything eagerly
            small = [b for b in BATCHES if b <= b0]
            ops = [("cap", b0) if o[0] in ("fused", "dropin", "dp") and rnd.random() < 0.6 else
                   ((o[0], rnd.choice(small)) if o[0] in ("fused", "dropin", "dp", "eval") else o) for o in ops]
            twins[0][0].engine.workspace(b0)
            caps[0] = bilinear_amd.CapturedTrainStep(twins[0][0], twins[0][1], b0, max_norm=1.0)
        first_bad = None
        for i, op in enumerate(ops):
            res = [apply(op, *tw, data, cap) for tw, cap in zip(twins, caps)]
            torch.cuda.synchronize()
            a, b = twins[0], twins[1]
            same = res[0] == res[1] and torch.equal(a[0].engine.params, b[0].engine.params) and \
                torch.equal(a[0].engine.bn_running, b[0].engine.bn_running)
            if a[1]._exp_avg is not None and b[1]._exp_avg is not None:
                same = same and torch.equal(a[1]._exp_avg_sq, b[1]._exp_avg_sq)
            if not same and first_bad is None:
                first_bad = (i, op, res)
        ok = first_bad is None
        bad += 0 if ok else 1
        print("sequence %d (%s twins, %s %d x %d, %d ops): %s" % (seq, twin, dtype, nb, width, nops, "twins bit-identical after every operation" if ok else
              "DIVERGED at op %d %s (results %s); ops so far: %s" % (first_bad[0], first_bad[1], first_bad[2], ops[:first_bad[0] + 1])),
              flush=True)
    print("%d sequences, %d diverged" % (nseq, bad), flush=True)
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 6, int(sys.argv[2]) if len(sys.argv) > 2 else 25,
                      sys.argv[3] if len(sys.argv) > 3 else "shadow") else 0)
